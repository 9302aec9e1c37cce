"""CPU tests of the host-side mirror of the reference surface: config keys, CLI commands/defaults, checkpoint
directory contract (rotation, `checkpoint` pointer, latest resolution)."""
import os
import numpy as np
import pytest
import yaml
from click.testing import CliRunner

from composer_amd import checkpoint as ckpt
from composer_amd import cli, config


def test_default_config_has_the_reference_keys_and_values():
    c = config.get(cli.get_default_config())
    ref = yaml.safe_load("""
window_size: 1024
embedding_size: 256
decoder_layers_count: 8
attention_head_count: 16
use_relative_attention: false
attention_dropout_rate: 0.1
residual_dropout_rate: 0.1
layer_normalization_epsilon: 0.00001
scale_attention: true
initializer_mean: 0
initializer_stddev: 0.02
use_layer_normalization: true
""")                                             # reference composer/default_config.yml:33-45
    assert dict(c.transformer.model) == ref
    assert c.transformer.train.batch_size == 1 and c.transformer.train.learning_rate == 0.001
    assert (c.dataset.time_step_increment, c.dataset.max_time_steps, c.dataset.velocity_bins) == (10, 100, 32)
    assert cli._vocab(c) == 390
    assert c.filepath == cli.get_default_config()


def test_config_objects_behave_like_mappings_and_like_attributes(tmp_path):
    import copy
    import pickle
    f = tmp_path / "two_docs.yml"
    f.write_text("a: {b: {c: 1}, l: [{x: 2}]}\nk: 1\n---\nk: 2\n---\n")       # later documents win; an empty document is skipped
    c = config.get(f)
    assert c.a.b.c == 1 and c["a"]["b"]["c"] == 1 and c.a.l[0].x == 2 and c.k == 2
    assert c.a.get("missing", 7) == 7 and c.get("a") is c.a
    c.a.b.new = {"deep": {"er": 3}}                      # assignment through either notation wraps nested mappings
    c["z"] = {"y": 4}
    assert c.a.b.new.deep.er == 3 and c.z.y == 4
    del c.z
    assert "z" not in c and not hasattr(c, "z")          # a missing key is an AttributeError, not a KeyError
    with pytest.raises(AttributeError):
        c.nope
    with pytest.raises(KeyError):
        c["nope"]
    assert "filepath" not in c and c.filepath == f       # where the file came from is not a configuration key
    for d in (copy.deepcopy(c), pickle.loads(pickle.dumps(c))):
        assert d == c and d.filepath == f and d.a.b.c == 1 and isinstance(d, config.ConfigInstance)
    assert yaml.safe_load(yaml.safe_dump(c.to_dict())) == c.to_dict() and type(c.to_dict()["a"]) is dict
    assert config.Dotdict({"a": 1}, b=2) == {"a": 1, "b": 2} and config.Dotdict().setdefault("q", {"r": 1}).r == 1


def test_cli_commands_and_defaults():
    r = CliRunner()
    out = r.invoke(cli.cli, ["--help"]).output
    for cmd in ("train", "evaluate", "generate", "make-config", "summary", "export-dataset"):
        assert cmd in out
    h = r.invoke(cli.cli, ["train", "--help"]).output
    for opt in ("--logdir", "--restoredir", "--config", "--epochs", "--max-files", "--save-freq-mode", "--save-freq",
                "--max-checkpoints", "--show-progress-bar"):
        assert opt in h
    p = {o.name: o.default for o in cli.train.params}
    assert p["epochs"] == 10 and p["save_frequency"] == 500 and p["max_checkpoints"] == 3 and p["logdir"] == "./output/logdir/"
    g = {o.name: o.default for o in cli.generate.params}
    assert g["prompt_length"] == 10 and g["generate_length"] == 1024 and g["temperature"] == 1.0
    # enum arguments are case-insensitive (click_utils.EnumType)
    assert cli.EnumType(cli.ModelType).convert("TRANSFORMER", None, None) == cli.ModelType.TRANSFORMER
    assert cli.EnumType(cli.ModelSaveFrequencyMode).convert("Global_Step", None, None) == cli.ModelSaveFrequencyMode.GLOBAL_STEP


def test_make_config_copies_the_default(tmp_path):
    r = CliRunner()
    dst = tmp_path / "c.yml"
    assert r.invoke(cli.cli, ["make-config", str(dst)]).exit_code == 0
    assert dst.read_text() == cli.get_default_config().read_text()


def test_missing_restoredir_config_exits_1(tmp_path):
    r = CliRunner()
    res = r.invoke(cli.cli, ["evaluate", "transformer", str(tmp_path), str(tmp_path / "nope")])
    assert res.exit_code == 1                      # cli.py:509-512


def test_checkpoint_manager_rotation_and_pointer(tmp_path):
    m = ckpt.CheckpointManager(tmp_path, max_to_keep=2)
    assert m.latest_checkpoint is None
    paths = [m.save({"model/w": np.full(3, i, np.float32), "optimizer/iter": np.int64(i)}, {"step": i, "epoch": 1}) for i in range(1, 5)]
    assert [os.path.basename(p) for p in paths] == ["ckpt-1", "ckpt-2", "ckpt-3", "ckpt-4"]
    assert sorted(f.name for f in tmp_path.glob("ckpt-*")) == ["ckpt-3.npz", "ckpt-4.npz"]       # max_to_keep
    txt = (tmp_path / "checkpoint").read_text()
    assert txt.startswith('model_checkpoint_path: "ckpt-4"') and 'all_model_checkpoint_paths: "ckpt-3"' in txt
    m2 = ckpt.CheckpointManager(tmp_path, max_to_keep=2)
    assert os.path.basename(m2.latest_checkpoint) == "ckpt-4"
    t, meta = ckpt.load(m2.latest_checkpoint)
    assert t["model/w"].tolist() == [4, 4, 4] and meta["step"] == 4 and meta["save_counter"] == 4
    assert os.path.basename(m2.save({"model/w": np.zeros(1, np.float32)}, {"step": 5, "epoch": 2})) == "ckpt-5"
    with pytest.raises(FileNotFoundError):
        ckpt.load(None)


def test_scalar_log_names(tmp_path):
    s = ckpt.ScalarLog(tmp_path / "train")
    s.scalar("loss", 1.5, 1); s.scalar("epoch_accuracy", 0.25, 2); s.close()
    lines = (tmp_path / "train" / "scalars.jsonl").read_text().strip().split("\n")
    assert '"tag": "loss"' in lines[0] and '"step": 2' in lines[1]


def test_bench_self_launches_n_ranks_and_fails_cleanly_without_gpus():
    """`python bench.py --gpus 2` (the driver's SCALE command shape) must start two ranks of itself under
    torch.distributed.run before touching the GPU; in this GPU-less container each rank then stops with a clear
    message and the parent forwards the launcher's non-zero status (no hang, no traceback from the parent)."""
    import subprocess, sys
    if __import__("torch").cuda.is_available():
        pytest.skip("GPU present: the ranks would run the real benchmark")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 0 needs GPU 0" in r.stderr and "rank 1 needs GPU 1" in r.stderr
    assert r.stdout.strip() == ""
