#!/usr/bin/env python3
"""oracle/run_tf_reference.py -- the ONE route from "parity unpinned" to "pinned" (SURVEY 8c).  TEST INFRASTRUCTURE: runs in the
build container only (it imports the reference from /root/reference), never ships, is never imported by the product.

The arithmetic of the path (composer/models/transformer.py:599-960) lives in TensorFlow, which is neither vendored nor
installable here, so the float64 oracle (oracle/transformer_oracle.py) is pinned only by a second, independent restatement
(oracle/torch_restatement.py).  The day `import tensorflow` succeeds in this container, this script settles it:

  1. builds the REAL `composer.models.transformer.Transformer` for every committed golden (tests/golden/transformer_g{A,B,C}.npz),
     with dropout 0 (the goldens' trajectories are dropout-free: the HIP path's counter-hash masks have no TF counterpart),
  2. assigns the golden's weights through the reference's own object graph -- `model.wte.weight` (transformer.py:117),
     `model.wpe.embeddings` (:673-679), `model.decoder_blocks[i].ln_1/ln_2.{gamma,beta}` (:551,563),
     `.attn.c_attn/.attn.c_proj.{weight,bias}` (:257-270; weight [hidden, filter], bias [1, filter], :189-190),
     `.mlp.c_fc/.mlp.c_proj` (:482-495), `model.ln_f.{gamma,beta}` (:694),
  3. diffs against the golden: logits of batch 0 (training=False, :696-833), the 10-step loss / accuracy trajectory of the
     train-loop body (:914-930: GradientTape, SparseCategoricalCrossentropy(from_logits=True), Keras Adam lr 1e-3), the parameters
     after step 3, and the greedy decode of cli.py:659-676 with argmax in place of tf.random.categorical (literal mode: the
     whole sequence re-fed; kv mode: model(x, past=presents)),
  4. writes the artefacts rows f2 / f4 could never observe -- a tf.train.Checkpoint of (step, epoch, optimizer, model) exactly as
     train() builds it (:887-891), a TFRecord through the reference's exporter (models/__init__.py:315-374) and a summary event file
     (:903, 933-951) -- into --out, and reads each back with the product's TensorFlow-free readers (composer_amd.tensorbundle /
     tfrecord / tbevents).  Commit what it writes under tests/golden/tf/ and the rows are pinned.

Exit status: 0 every comparison within tolerance; 1 a mismatch (printed); 3 TensorFlow not importable (today's result).
    python oracle/run_tf_reference.py [--out DIR] [--golden gA,gB,gC] [--tol 2e-4]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("COMPOSER_REFERENCE", "/root/reference")
DECODE_SCALE_KEY = "decode_scale"


def import_tensorflow():
    try:
        import tensorflow as tf          # noqa: F401
        return tf
    except Exception as e:               # ImportError, or a broken wheel
        print("run_tf_reference: TensorFlow not importable (%s: %s) -- parity stays unpinned" % (type(e).__name__, e))
        return None


def golden_params(g):
    return {k[len("param:"):]: np.asarray(g[k], np.float32) for k in g.files if k.startswith("param:")}


def build_reference_model(tf, cfg, params):
    """cfg = (V, E, H, L, W, T, B) as stored in the golden.  Returns the reference model with the golden's weights."""
    sys.path.insert(0, REFERENCE)
    from composer.models.transformer import Transformer
    V, E, H, L, W, T, B = [int(v) for v in cfg]
    model = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0)      # positional order of :610-614
    model(tf.zeros((1, 2), tf.int32))                                                               # creates every variable
    assign_parameters(model, params, L)
    return model


def reference_variable_paths(L):
    """checkpoint name of the product (include/composer_hip.h) -> attribute path from the reference model object, e.g.
    'decoder_blocks/3/attn/c_attn/weight' -> ('decoder_blocks', 3, 'attn', 'c_attn', 'weight').  Pure: no TensorFlow needed
    (tests/test_oracle.py checks every attribute against the reference's source)."""
    out = {"wte/weight": ("wte", "weight"), "wpe/embeddings": ("wpe", "embeddings"),
           "ln_f/gamma": ("ln_f", "gamma"), "ln_f/beta": ("ln_f", "beta")}
    for i in range(L):
        p, b = "decoder_blocks/%d/" % i, ("decoder_blocks", i)
        for ln in ("ln_1", "ln_2"):
            out[p + ln + "/gamma"], out[p + ln + "/beta"] = b + (ln, "gamma"), b + (ln, "beta")
        for sub, conv in (("attn", "c_attn"), ("attn", "c_proj"), ("mlp", "c_fc"), ("mlp", "c_proj")):
            out[p + sub + "/" + conv + "/weight"] = b + (sub, conv, "weight")
            out[p + sub + "/" + conv + "/bias"] = b + (sub, conv, "bias")
    return out


def reference_variables(model, L):
    """checkpoint name of the product -> the reference's tf.Variable."""
    out = {}
    for name, path in reference_variable_paths(L).items():
        obj = model
        for step in path:
            obj = obj[step] if isinstance(step, int) else getattr(obj, step)
        out[name] = obj
    return out


def assign_parameters(model, params, L):
    ref = reference_variables(model, L)
    missing = sorted(set(params) - set(ref)), sorted(set(ref) - set(params))
    assert not missing[0] and not missing[1], "name mismatch between the golden and the reference's object graph: %r" % (missing,)
    for name, var in ref.items():
        val = np.asarray(params[name], np.float32).reshape(var.shape)           # (Conv1D bias is [1, filter] on both sides)
        var.assign(val)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def check_golden(tf, name, tol, report):
    g = np.load(os.path.join(ROOT, "tests", "golden", "transformer_%s.npz" % name))
    cfg = g["cfg"]
    V, E, H, L, W, T, B = [int(v) for v in cfg]
    params = golden_params(g)
    ok = True

    def note(what, err, limit):
        nonlocal ok
        good = err <= limit
        ok &= good
        report.append("%s %-34s err %.3e (limit %.1e) %s" % (name, what, err, limit, "ok" if good else "MISMATCH"))

    model = build_reference_model(tf, cfg, params)
    logits, _ = model(tf.constant(g["x"][0], tf.int32), training=False)
    note("logits of batch 0", rel(logits.numpy(), g["logits0"]), tol)

    # the train-loop body, transformer.py:914-930
    optimizer = tf.keras.optimizers.Adam(learning_rate=float(g["lr"]))
    loss_object = tf.keras.losses.SparseCategoricalCrossentropy(from_logits=True)
    losses, accs = [], []
    for s in range(g["x"].shape[0]):
        x, y = tf.constant(g["x"][s], tf.int32), tf.constant(g["y"][s], tf.int32)
        with tf.GradientTape() as tape:
            predictions, _ = model(x, training=True)
            loss = loss_object(y_true=y, y_pred=predictions)
        grads = tape.gradient(loss, model.trainable_variables)
        optimizer.apply_gradients(zip(grads, model.trainable_variables))
        acc = tf.reduce_mean(tf.cast(tf.equal(tf.cast(tf.argmax(predictions, axis=-1), tf.int32), y), tf.float32))
        losses.append(float(loss)); accs.append(float(acc))
        if s == 2:
            ref = reference_variables(model, L)
            for k in ref:
                if "param3:" + k in g.files:
                    note("param after step 3: " + k[-24:], rel(ref[k].numpy().reshape(g["param3:" + k].shape), g["param3:" + k]), 5 * tol)
                else:
                    nrm = float(np.sqrt((ref[k].numpy().astype(np.float64) ** 2).sum()))
                    note("|param| after step 3: " + k[-22:], abs(nrm - float(g["param3norm:" + k])) / (float(g["param3norm:" + k]) + 1e-30), 5 * tol)
    note("10-step loss trajectory", rel(losses, g["losses"]), tol)
    note("10-step accuracy trajectory", float(np.abs(np.array(accs) - g["accs"]).max()), 1e-6)

    # greedy decode of cli.py:659-676 on the decode fixture's weights (truncated-normal tensors x decode_scale, float32)
    sys.path.insert(0, ROOT)
    from oracle import transformer_oracle as O
    kinds = {n: k for n, _, k in O.param_specs(V, E, W, L)}
    scale = np.float32(g[DECODE_SCALE_KEY])
    dparams = {k: (v * scale if kinds[k] == "normal" else v).astype(np.float32) for k, v in params.items()}
    assign_parameters(model, dparams, L)
    prompt = [int(t) for t in g["prompt"]]
    n = len(g["greedy_literal"])
    ids = list(prompt)                                                   # literal: `past` never fed back, x = the new id only (:675)
    x = tf.constant([prompt], tf.int32)
    lit = []
    for _ in range(n):
        z = model(x)[0]
        t = int(tf.argmax(z[0, -1]).numpy())
        lit.append(t)
        x = tf.constant([[t]], tf.int32)
    note("greedy ids, literal mode", float(np.mean(np.array(lit) != g["greedy_literal"])), 0.0)
    kv, past = [], None
    x = tf.constant([prompt], tf.int32)
    for _ in range(n):
        z, past = model(x, past=past)[:2]
        t = int(tf.argmax(z[0, -1]).numpy())
        kv.append(t)
        ids.append(t)
        x = tf.constant([ids], tf.int32)                                 # with `past` the call keeps the last token only (:735-737)
    note("greedy ids, kv mode", float(np.mean(np.array(kv) != g["greedy_kv"])), 0.0)
    return ok, model, g


def write_and_read_back(tf, model, g, out_dir, report):
    """f2 / f4: TF-written checkpoint, TFRecord and event file, read back with the product's readers."""
    from composer_amd import tensorbundle, tfrecord, tbevents
    os.makedirs(out_dir, exist_ok=True)
    ok = True
    V, E, H, L, W, T, B = [int(v) for v in g["cfg"]]
    # -- checkpoint, as Transformer.train builds it (:887-891)
    optimizer = tf.keras.optimizers.Adam(learning_rate=1e-3)
    with tf.GradientTape() as tape:
        loss = tf.reduce_mean(model(tf.constant(g["x"][0], tf.int32), training=True)[0])
    optimizer.apply_gradients(zip(tape.gradient(loss, model.trainable_variables), model.trainable_variables))   # creates the slots
    checkpoint = tf.train.Checkpoint(step=tf.Variable(1), epoch=tf.Variable(1), optimizer=optimizer, model=model)
    manager = tf.train.CheckpointManager(checkpoint, os.path.join(out_dir, "ckpt"), max_to_keep=1)
    prefix = manager.save()
    bundle = tensorbundle.read_bundle(prefix)
    tensors, meta = tensorbundle.state_from_bundle(bundle)
    ref = reference_variables(model, L)
    worst = max(rel(np.asarray(tensors["model/" + k]).reshape(v.shape), v.numpy()) for k, v in ref.items())
    for slot in ("m", "v"):                     # the Adam slots train() would restore (:896-899)
        assert all("optimizer/%s/%s" % (slot, k) in tensors for k in ref), "Adam slot names differ from the product's checkpoint keys"
    assert meta.get("step") == 1 and meta.get("epoch") == 1, meta
    report.append("f2 checkpoint read back: %d tensors, worst difference %.1e, keys e.g. %s" % (len(ref), worst, sorted(bundle)[:3]))
    ok &= worst == 0.0
    # -- TFRecord through the reference's exporter (models/__init__.py:315-374 reads what cli.py's export-dataset writes)
    sys.path.insert(0, REFERENCE)
    rec = os.path.join(out_dir, "dataset.tfrecord")
    with tf.io.TFRecordWriter(rec) as w:
        hdr = tf.train.Example(features=tf.train.Features(feature={
            "model_type": tf.train.Feature(bytes_list=tf.train.BytesList(value=[b"transformer"])),
            "batch_size": tf.train.Feature(int64_list=tf.train.Int64List(value=[B])),
            "window_size": tf.train.Feature(int64_list=tf.train.Int64List(value=[T]))}))
        w.write(hdr.SerializeToString())
        for s in range(g["x"].shape[0]):
            ex = tf.train.Example(features=tf.train.Features(feature={
                "x": tf.train.Feature(bytes_list=tf.train.BytesList(value=[tf.io.serialize_tensor(tf.constant(g["x"][s], tf.int32)).numpy()])),
                "y": tf.train.Feature(bytes_list=tf.train.BytesList(value=[tf.io.serialize_tensor(tf.constant(g["y"][s], tf.int32)).numpy()]))}))
            w.write(ex.SerializeToString())
    ds, header = tfrecord.load_tfrecord_dataset(rec, shuffle=False)
    got = list(ds)
    same = len(got) == g["x"].shape[0] and all(np.array_equal(a, g["x"][i]) and np.array_equal(b, g["y"][i]) for i, (a, b) in enumerate(got))
    report.append("f4 TFRecord read back: header %r, %d batches, identical %s" % (header, len(got), same))
    ok &= same
    # -- summary events (:903, 933-951)
    ev_dir = os.path.join(out_dir, "train")
    writer = tf.summary.create_file_writer(ev_dir)
    with writer.as_default():
        for s in range(5):
            tf.summary.scalar("loss", float(g["losses"][s]), step=s + 1)
            tf.summary.scalar("accuracy", float(g["accs"][s]), step=s + 1)
    writer.flush()
    files = [os.path.join(ev_dir, f) for f in os.listdir(ev_dir)]
    version, scalars = tbevents.read_scalars(files[0])
    want = [("loss", s + 1, float(np.float32(g["losses"][s]))) for s in range(5)]
    got_loss = [(t, st, v) for t, st, v, _ in scalars if t == "loss"]
    same = len(got_loss) == 5 and all(a[0] == b[0] and a[1] == b[1] and abs(a[2] - b[2]) < 1e-6 for a, b in zip(got_loss, want))
    report.append("f4 event file read back: version %r, %d scalars, loss series identical %s" % (version, len(scalars), same))
    ok &= same
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "tf"))
    ap.add_argument("--golden", default="gA,gB,gC")
    ap.add_argument("--tol", type=float, default=2e-4, help="float32 TensorFlow against the float64 oracle")
    args = ap.parse_args(argv)
    tf = import_tensorflow()
    if tf is None:
        return 3
    if not os.path.isdir(os.path.join(REFERENCE, "composer")):
        print("run_tf_reference: the reference is not at %s" % REFERENCE)
        return 3
    report, ok, last = [], True, None
    for name in args.golden.split(","):
        good, model, g = check_golden(tf, name, args.tol, report)
        ok &= good
        last = (model, g)
    ok &= write_and_read_back(tf, last[0], last[1], args.out, report)
    print("\n".join(report))
    print("run_tf_reference: %s" % ("every comparison within tolerance -- the oracle is PINNED to the TensorFlow reference; commit %s"
                                    % args.out if ok else "MISMATCH (see above)"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
