"""Host-side mirror of the reference's `composer.models.Transformer` (reference
composer/models/transformer.py:599-960) over libcomposer_hip.so.

Same constructor arguments (transformer.py:610-614), same call / train / evaluate / load_from_checkpoint
surface (cli.py:579-589, 606-615, 635-676).  All arithmetic happens in hand-written HIP kernels on an
MI355X; this file only moves integers in and numbers out.  There is no CPU path.
"""
import ctypes as C
import enum
import logging
import math
import os
import time
from pathlib import Path

import numpy as np

from . import _lib
from . import checkpoint as ckpt


class ModelSaveFrequencyMode(enum.Enum):
    """reference composer/models/__init__.py:92-107"""
    EPOCH = 'epoch'
    GLOBAL_STEP = 'global_step'


def _truncated_normal(rng, shape, mean, stddev):
    """tf.keras.initializers.TruncatedNormal: resample outside +-2 sigma (transformer.py:115,188,670-673)."""
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (mean + stddev * a).astype(np.float32)


class Presents:
    """Lazy `presents` tuple (transformer.py:797-806,820-821): L tensors [2,B,H,T,D] fetched from the saved
    c_attn activations only when indexed (the CLI never reads them)."""

    def __init__(self, model, B, T, generation):
        self._m, self._B, self._T, self._gen = model, B, T, generation

    def __len__(self):
        return self._m.decoder_layers_count

    def __getitem__(self, i):
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._m._fetch_present(i, self._B, self._T, self._gen)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Transformer:
    def __init__(self, vocab_size, embedding_size, window_size, decoder_layers_count,
                 attention_head_count, use_relative_attention=False, initializer_mean=0,
                 initializer_stddev=0.02, attention_dropout_rate=0.1, residual_dropout_rate=0.1,
                 layer_normalization_epsilon=1e-5, scale=True, use_layer_normalization=True,
                 output_hidden_states=False, output_attention_weights=False, *,
                 dtype='bf16', seed=0, max_batch=1, max_seq=None, device=None, ctx=None):
        if embedding_size % attention_head_count != 0:
            raise AssertionError('hidden size must be a multiple of the attention head count')   # transformer.py:255
        if use_relative_attention:
            # the reference path is broken (Attention.build reads an undefined self.depth, transformer.py:285)
            raise NotImplementedError('use_relative_attention is not supported (broken in the reference too)')
        self.output_hidden_states = bool(output_hidden_states)
        self.output_attention_weights = bool(output_attention_weights)
        self.vocab_size = vocab_size
        self.embedding_size = embedding_size
        self.window_size = window_size
        self.decoder_layers_count = decoder_layers_count
        self.attention_head_count = attention_head_count
        self.use_layer_normalization = use_layer_normalization
        self.initializer_mean = initializer_mean
        self.initializer_stddev = initializer_stddev
        self.dtype = {'fp32': _lib.CMP_FP32, 'float32': _lib.CMP_FP32, 'bf16': _lib.CMP_BF16,
                      'bfloat16': _lib.CMP_BF16}[str(dtype)]
        self.seed = int(seed)
        self._lib = _lib.load()
        _lib.require_gpu()
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
        self._own_ctx = ctx is None
        if ctx is None:
            h = C.c_void_p()
            _lib.check(self._lib.cmp_ctx_create(int(device), C.byref(h)), 'cmp_ctx_create')
            ctx = h
        self._ctx = ctx
        cfg = _lib.ModelCfg(vocab_size, embedding_size, window_size, decoder_layers_count, attention_head_count,
                            float(layer_normalization_epsilon), int(bool(scale)), int(bool(use_layer_normalization)),
                            float(attention_dropout_rate), float(residual_dropout_rate), self.dtype,
                            int(max_batch), int(max_seq or window_size), self.seed)
        h = C.c_void_p()
        rc = self._lib.cmp_model_create(self._ctx, C.byref(cfg), C.byref(h))
        if rc != 0:
            msg = _lib.last_error()
            if self._own_ctx:                       # a refused configuration must not leave its context (three streams) behind
                self._lib.cmp_ctx_destroy(self._ctx)
            self._ctx = None
            raise _lib.HipLibraryError("cmp_model_create failed (status %d): %s" % (rc, msg))
        self._h = h
        self._specs = self._param_specs()
        self._learning_rate = 1e-3
        self._dp = None          # (rank, nranks) once init_data_parallel() ran
        self.initialize_parameters(self.seed)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, '_h', None):
            self._lib.cmp_model_destroy(self._h)
            self._h = None
            if self._own_ctx and self._ctx:
                self._lib.cmp_ctx_destroy(self._ctx)
            self._ctx = None
        elif getattr(self, '_ctx', None) and getattr(self, '_own_ctx', False):     # construction stopped after the context
            self._lib.cmp_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _param_specs(self):
        n = C.c_int()
        _lib.check(self._lib.cmp_param_count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name, rank, shape, numel = C.c_char_p(), C.c_int(), (C.c_int64 * 4)(), C.c_int64()
            _lib.check(self._lib.cmp_param_info(self._h, i, C.byref(name), C.byref(rank), C.byref(shape), C.byref(numel)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(rank.value)), int(numel.value)))
        return out

    @property
    def parameter_names(self):
        return [n for n, _, _ in self._specs]

    def parameter_shape(self, name):
        return {n: s for n, s, _ in self._specs}[name]

    def get_parameter(self, name, kind=_lib.KIND_VALUE):
        shape = self.parameter_shape(name)
        a = np.empty(shape, np.float32)
        _lib.check(self._lib.cmp_param_get(self._h, name.encode(), kind, a.ctypes.data_as(C.c_void_p), a.size), 'cmp_param_get')
        return a

    def set_parameter(self, name, value, kind=_lib.KIND_VALUE):
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(self.parameter_shape(name)))
        _lib.check(self._lib.cmp_param_set(self._h, name.encode(), kind, a.ctypes.data_as(C.c_void_p), a.size), 'cmp_param_set')

    def get_weights(self):
        return {n: self.get_parameter(n) for n in self.parameter_names}

    def set_weights(self, weights):
        for n, v in weights.items():
            self.set_parameter(n, v)

    def initialize_parameters(self, seed=0):
        """TruncatedNormal(mean, stddev) for wte/wpe/Conv1D weights, zeros for biases/beta, ones for gamma
        (transformer.py:115,188-190,670-673; Keras LayerNormalization defaults)."""
        rng = np.random.default_rng(seed)
        for name, shape, _ in self._specs:
            if name.endswith('gamma'):
                v = np.ones(shape, np.float32)
            elif name.endswith(('beta', 'bias')):
                v = np.zeros(shape, np.float32)
            else:
                v = _truncated_normal(rng, shape, self.initializer_mean, self.initializer_stddev)
            self.set_parameter(name, v)

    @property
    def iterations(self):
        v = C.c_int64()
        _lib.check(self._lib.cmp_adam_iter_get(self._h, C.byref(v)))
        return int(v.value)

    @iterations.setter
    def iterations(self, v):
        _lib.check(self._lib.cmp_adam_iter_set(self._h, int(v)))

    # ------------------------------------------------------------------ Keras-surface no-ops used by the CLI
    def compile(self, learning_rate):            # transformer.py:835-844
        self._learning_rate = float(learning_rate)

    def build(self, input_shape=None):           # cli.py:607,640
        return None

    def reset_states(self):                      # cli.py:662
        return None

    def summary(self, print_fn=print):           # cli.py:436-440
        total = 0
        print_fn('Model: "transformer"')
        for n, s, k in self._specs:
            print_fn('%-48s %-16s %d' % (n, s, k))
            total += k
        print_fn('Total params: {:,}'.format(total))
        return total

    # ------------------------------------------------------------------ data parallel
    def init_data_parallel(self, rank, world_size, unique_id, gemm_cus=None):
        """One rank per GPU; `unique_id` = 128 bytes from rank 0's `new_unique_id()` shared by the launcher.
        gemm_cus: CUs the persistent GEMM kernels may occupy while gradients are all-reduced (None/0 = all 256)."""
        _lib.check_single_runtime()      # a second RCCL mapped beside the one the library is bound to: fail here, with the paths
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self._lib.cmp_dp_init(self._ctx, int(rank), int(world_size), buf), 'cmp_dp_init')
        if gemm_cus:
            _lib.check(self._lib.cmp_dp_set_gemm_cus(self._ctx, int(gemm_cus)), 'cmp_dp_set_gemm_cus')
        self._dp = (int(rank), int(world_size))

    def init_data_parallel_exchange(self, rank, world_size, all_reduce):
        """The data-parallel step over the caller's transport (cmp_dp_init_exchange): `all_reduce(dev_ptr, count, hip_stream)` must
        leave the sum over all ranks of `count` float32 at device address `dev_ptr`, ordered on `hip_stream`; it is called once per
        gradient bucket (and once for the 3-float metrics message), on every rank in the same order.  An exception it raises fails
        the train step (HipLibraryError) and is kept in `self.exchange_error`."""
        self.exchange_error = None

        def _cb(user, ptr, count, stream):
            try:
                all_reduce(int(ptr), int(count), int(stream or 0))
                return 0
            except BaseException as e:          # never unwind through the C frames
                self.exchange_error = e
                return 1
        self._xcb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(_cb)     # kept alive with the model
        _lib.check(self._lib.cmp_dp_init_exchange(self._ctx, int(rank), int(world_size), C.cast(self._xcb, C.c_void_p), None),
                   'cmp_dp_init_exchange')
        self._dp = (int(rank), int(world_size))

    def set_mask_rank(self, rank):
        """Rank folded into the dropout seed (seed ^ mix32(rank)); init_data_parallel sets it to the communicator rank."""
        _lib.check(self._lib.cmp_dp_set_mask_rank(self._ctx, int(rank)), 'cmp_dp_set_mask_rank')

    def all_reduce_sum(self, values):
        """Sum of a small float vector over the data-parallel ranks (RCCL, cmp_dp_allreduce_test)."""
        a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
        _lib.check(self._lib.cmp_dp_allreduce_test(self._ctx, a.ctypes.data_as(C.c_void_p), a.size), 'cmp_dp_allreduce_test')
        return a

    def all_reduce_pattern(self, reps=20):
        """The gradient exchange of one train step alone (cmp_dp_allreduce_pattern): the step's own bucket pattern back to back on the
        communication stream.  Returns ms per repetition, bytes and messages per repetition.  Collective: every rank calls it."""
        ms, nbytes, msgs = C.c_double(), C.c_int64(), C.c_int()
        _lib.check(self._lib.cmp_dp_allreduce_pattern(self._h, int(reps), C.byref(ms), C.byref(nbytes), C.byref(msgs)),
                   'cmp_dp_allreduce_pattern')
        return {"ms": ms.value / reps, "bytes": nbytes.value, "messages": msgs.value, "reps": int(reps)}

    def dp_stats(self, reset=False):
        """Gradient-exchange telemetry since the last reset (cmp_dp_stats): steps, the communication time per step that the
        backward pass did not hide (ms), bytes and all-reduce calls per step.  Zeros without a communicator."""
        steps, ms, nbytes, msgs = C.c_int64(), C.c_double(), C.c_int64(), C.c_int()
        _lib.check(self._lib.cmp_dp_stats(self._h, int(bool(reset)), C.byref(steps), C.byref(ms), C.byref(nbytes), C.byref(msgs)),
                   'cmp_dp_stats')
        return {"steps": steps.value, "exposed_ms": ms.value / steps.value if steps.value else 0.0,
                "exposed_ms_total": ms.value, "bytes": nbytes.value, "buckets": msgs.value}

    @staticmethod
    def new_unique_id():
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().cmp_dp_unique_id(buf), 'cmp_dp_unique_id')
        return buf.raw

    # ------------------------------------------------------------------ forward / steps
    @staticmethod
    def _ids(x):
        a = np.asarray(x)
        if a.dtype != np.int32 or not a.flags.c_contiguous:         # (an int32 C-ordered batch is handed over as it is: no copy per step)
            a = np.ascontiguousarray(a.astype(np.int32))
        if a.ndim == 1:
            a = a[None]
        if a.ndim != 2:
            raise ValueError('expected an int tensor with shape [batch, sequence]')
        return a

    def _check_ids(self, a):
        # one pass: a negative id is a huge unsigned one
        if a.size and int(a.view(np.uint32).max()) >= self.vocab_size:
            raise ValueError('token id outside [0, %d)' % self.vocab_size)

    def __call__(self, inputs, past=None, attention_mask=None, token_type_ids=None, position_ids=None,
                 input_embeddings=None, use_cache=True, training=False):
        """Transformer.call (transformer.py:696-833) -> (logits [B,T,V] float32, presents[, all_hidden_states][, all_attentions]).

        `past` = an earlier call's presents (L tensors [2,B,H,Tp,D], or the lazy Presents object): only the last input
        token is used (:735-737), it sits at position Tp (:760-770), its keys/values are appended to `past` (:423-426)
        and the returned presents hold Tp+1 positions.  `training=True` (:916-917) applies dropout with the masks a train
        step at the current optimizer iteration would draw; together with `past` the attention-probability mask is the new
        token's row of the mask over all Tp+1 positions."""
        if input_embeddings is not None:
            # (it cannot work in the reference either: transformer.py:758 casts inputs=None)
            raise NotImplementedError('input_embeddings is never used by the CLI path')
        x = self._ids(inputs)
        past_len, past_ptrs, keep = 0, None, []
        if past is not None:
            x = np.ascontiguousarray(x[:, -1:])                                  # transformer.py:735-737
            past = list(past)
            if len(past) != self.decoder_layers_count:
                raise ValueError('past must hold one tensor per decoder block')
            H, D = self.attention_head_count, self.embedding_size // self.attention_head_count
            keep = [np.ascontiguousarray(np.asarray(p, dtype=np.float32)) for p in past]
            past_len = int(keep[0].shape[-2])
            for p in keep:
                if p.shape != (2, x.shape[0], H, past_len, D):
                    raise ValueError('past tensors must be [2, batch, heads, past_len, head_size]; got %s' % (p.shape,))
            past_ptrs = (C.c_void_p * len(keep))(*[p.ctypes.data for p in keep])
        self._check_ids(x)
        B, T = x.shape
        if past_len + T > self.window_size:
            raise IndexError('position %d outside the wpe table (window_size %d, transformer.py:675-679,786)'
                             % (past_len + T - 1, self.window_size))
        logits = np.empty((B, T, self.vocab_size), np.float32)
        pos = typ = None
        if token_type_ids is not None:                                           # :787-791: a second wte row per token
            typ = self._ids(token_type_ids)
            if past is not None:
                typ = typ[:, -1:]                                                # :741-742
            typ = np.ascontiguousarray(np.broadcast_to(typ.reshape(-1, typ.shape[-1]), (B, T)))
            self._check_ids(typ)
        if position_ids is not None:                                             # :770-773, 784, 786: rows of wpe, [1,T] or [B,T]
            pos = self._ids(position_ids)
            pos = np.ascontiguousarray(np.broadcast_to(pos.reshape(-1, pos.shape[-1]), (B, T)))
            if pos.size and (pos.min() < 0 or pos.max() >= self.window_size):
                raise IndexError('position id outside the wpe table (window_size %d, transformer.py:675-679,786)' % self.window_size)
        amask = None
        if attention_mask is not None:                                           # :774-779, 356-358: [B, past + new keys], 1 = attend
            raw = np.asarray(attention_mask)
            # the reference casts the mask to float32 and adds (1 - mask) * -1e4 (:774-779): a fractional entry is a soft mask there.
            # The C ABI carries 0 / 1 integers, so anything else is refused rather than truncated to "masked".
            if raw.size and not np.all((raw == 0) | (raw == 1)):
                raise ValueError('attention_mask entries must be 0 or 1 (soft masks are not supported by the HIP path)')
            amask = np.ascontiguousarray(raw.astype(np.int32))
            if amask.shape != (B, past_len + T):
                raise ValueError('attention_mask must be [batch, past_len + sequence] = %s; got %s' % ((B, past_len + T), amask.shape))
        attentions, att_ptrs = None, None
        if self.output_attention_weights:                                        # :360-369, 808-809: [B, H, new queries, all keys] per block
            H = self.attention_head_count
            attentions = [np.empty((B, H, T, past_len + T), np.float32) for _ in range(self.decoder_layers_count)]
            att_ptrs = (C.c_void_p * len(attentions))(*[a.ctypes.data for a in attentions])
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.cmp_forward_ex(self._h, x.ctypes.data_as(C.c_void_p), B, T, past_len, past_ptrs, int(bool(training)),
                                            ptr(pos), ptr(typ), ptr(amask), att_ptrs, logits.ctypes.data_as(C.c_void_p)),
                   'cmp_forward_ex')
        gen = C.c_int64()
        _lib.check(self._lib.cmp_forward_generation(self._h, C.byref(gen)), 'cmp_forward_generation')
        outputs = (logits,)
        if use_cache is True:
            outputs += (Presents(self, B, past_len + T, int(gen.value)),)
        if self.output_hidden_states:
            # transformer.py:800-816, 824-825: the input of every decoder block, then the ln_f output -- L + 1 tensors [B, T, E]
            hidden = []
            for i in range(self.decoder_layers_count + 1):
                h = np.empty((B, T, self.embedding_size), np.float32)
                _lib.check(self._lib.cmp_hidden_get_at(self._h, i, B, T, int(gen.value), h.ctypes.data_as(C.c_void_p)),
                           'cmp_hidden_get_at')
                hidden.append(h)
            outputs += (tuple(hidden),)
        if attentions is not None:
            outputs += (tuple(attentions),)                                      # :827-831
        return outputs

    def _fetch_present(self, layer, B, T, generation):
        """presents[layer] = stack([key, value]) [2,B,H,T,D] of the forward pass that produced this Presents object.  Valid
        until the next forward / train step of the model: a read after that raises (never another pass's tensors)."""
        H = self.attention_head_count
        out = np.empty((2, B, H, T, self.embedding_size // H), np.float32)
        _lib.check(self._lib.cmp_present_get_at(self._h, int(layer), B, T, int(generation), out.ctypes.data_as(C.c_void_p)),
                   'cmp_present_get_at')
        return out

    def train_step(self, x, y, learning_rate=None, sync=True):
        """One iteration of transformer.py:914-930: returns (loss, accuracy) of this rank's batch."""
        x, y = self._ids(x), self._ids(y)
        self._check_ids(x); self._check_ids(y)
        B, T = x.shape
        lr = self._learning_rate if learning_rate is None else float(learning_rate)
        loss, acc = C.c_float(), C.c_float()
        _lib.check(self._lib.cmp_train_step(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), B, T, lr,
                                            C.byref(loss) if sync else None, C.byref(acc) if sync else None), 'cmp_train_step')
        return (loss.value, acc.value) if sync else None

    def train_step_device(self, x_ptr, y_ptr, B, T, learning_rate=None):
        """x_ptr / y_ptr: HIP device pointers to int32 [B,T] (e.g. torch tensor .data_ptr()); no host sync."""
        lr = self._learning_rate if learning_rate is None else float(learning_rate)
        _lib.check(self._lib.cmp_train_step_dev(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), B, T, lr), 'cmp_train_step_dev')

    def train_step_async(self, x, y, learning_rate=None):
        """Submits one train step without waiting for it (ids staged in pinned memory, uploaded on a copy stream behind the
        previous step); returns a ticket for step_metrics().  At most 3 steps are in flight."""
        x, y = self._ids(x), self._ids(y)
        self._check_ids(x); self._check_ids(y)
        B, T = x.shape
        lr = self._learning_rate if learning_rate is None else float(learning_rate)
        ticket = C.c_int64()
        _lib.check(self._lib.cmp_train_step_async(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), B, T, lr,
                                                  C.byref(ticket)), 'cmp_train_step_async')
        return int(ticket.value)

    def step_metrics(self, ticket):
        """(loss, accuracy) of the step `ticket` was issued for; blocks until that step (not later ones) has finished."""
        loss, acc = C.c_float(), C.c_float()
        _lib.check(self._lib.cmp_train_metrics_wait(self._h, int(ticket), C.byref(loss), C.byref(acc)), 'cmp_train_metrics_wait')
        return loss.value, acc.value

    def last_metrics(self):
        loss, acc = C.c_float(), C.c_float()
        _lib.check(self._lib.cmp_train_metrics(self._h, C.byref(loss), C.byref(acc)), 'cmp_train_metrics')
        return loss.value, acc.value

    def synchronize(self):
        _lib.check(self._lib.cmp_sync(self._ctx), 'cmp_sync')

    @property
    def stream(self):
        return self._lib.cmp_ctx_stream(self._ctx)

    def loss_and_grads(self, x, y):
        """forward(training=True)+backward without the optimizer; gradients via get_parameter(name, KIND_GRAD)."""
        x, y = self._ids(x), self._ids(y)
        self._check_ids(x); self._check_ids(y)
        B, T = x.shape
        loss, acc = C.c_float(), C.c_float()
        _lib.check(self._lib.cmp_loss_and_grads(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), B, T,
                                                C.byref(loss), C.byref(acc)), 'cmp_loss_and_grads')
        return loss.value, acc.value

    def evaluate(self, dataset, verbose=0):
        """model.evaluate(ds) (cli.py:613): mean sparse-CE and token accuracy over all batches."""
        tot, cor, cnt = 0.0, 0, 0
        for x, y in dataset:
            x, y = self._ids(x), self._ids(y)
            self._check_ids(x); self._check_ids(y)
            B, T = x.shape
            ls, c, n = C.c_double(), C.c_int64(), C.c_int64()
            _lib.check(self._lib.cmp_eval_step(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), B, T,
                                               C.byref(ls), C.byref(c), C.byref(n)), 'cmp_eval_step')
            tot += ls.value; cor += c.value; cnt += n.value
        if cnt == 0:
            return float('nan'), float('nan')
        return tot / cnt, cor / cnt

    # ------------------------------------------------------------------ decode (cli.py:659-676)
    def generate(self, prompt_ids, length, temperature=1.0, mode='kv', seed=None):
        """Returns `length` generated ids.  mode 'literal' restates cli.py:663-676 as written (no `past`),
        mode 'kv' is model(x, past=presents).  temperature <= 0 -> greedy argmax."""
        p = np.ascontiguousarray(np.asarray(prompt_ids, dtype=np.int32).reshape(-1))
        m = {'literal': _lib.DECODE_LITERAL, 'reference-literal': _lib.DECODE_LITERAL, 'kv': _lib.DECODE_KV,
             'kv-cache': _lib.DECODE_KV}[mode]
        if m == _lib.DECODE_KV and len(p) + length - 1 > self.window_size:
            raise IndexError('prompt_len + length - 1 = %d exceeds window_size %d (wpe rows, transformer.py:675-679,786)'
                             % (len(p) + length - 1, self.window_size))
        _lib.check(self._lib.cmp_decode_begin(self._h, p.ctypes.data_as(C.c_void_p), len(p), m, float(temperature),
                                              int(self.seed if seed is None else seed)), 'cmp_decode_begin')
        out = np.empty(length, np.int32)
        _lib.check(self._lib.cmp_decode_steps(self._h, int(length), out.ctypes.data_as(C.c_void_p)), 'cmp_decode_steps')
        return out

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        sd = {}
        for n in self.parameter_names:
            sd['model/' + n] = self.get_parameter(n)
            sd['optimizer/m/' + n] = self.get_parameter(n, _lib.KIND_ADAM_M)
            sd['optimizer/v/' + n] = self.get_parameter(n, _lib.KIND_ADAM_V)
        sd['optimizer/iter'] = np.int64(self.iterations)
        return sd

    def load_state_dict(self, sd, expect_partial=False):
        for n in self.parameter_names:
            self.set_parameter(n, sd['model/' + n])
            if 'optimizer/m/' + n in sd:
                self.set_parameter(n, sd['optimizer/m/' + n], _lib.KIND_ADAM_M)
                self.set_parameter(n, sd['optimizer/v/' + n], _lib.KIND_ADAM_V)
            elif not expect_partial:
                raise KeyError('optimizer slot for ' + n)
        if 'optimizer/iter' in sd:
            self.iterations = int(sd['optimizer/iter'])

    def load_from_checkpoint(self, restoredir):
        """BaseModel.load_from_checkpoint (reference composer/models/__init__.py:66-90): restores the latest
        checkpoint in `restoredir` (model only, expect_partial); logs and exits(1) on failure."""
        try:
            mgr = ckpt.CheckpointManager(restoredir, max_to_keep=None)
            if mgr.latest_checkpoint is None:
                raise FileNotFoundError('no checkpoint in ' + str(restoredir))
            sd, _meta = ckpt.load(mgr.latest_checkpoint)
            self.load_state_dict(sd, expect_partial=True)
            logging.info('Model restored from \'{}\'.'.format(mgr.latest_checkpoint))
        except Exception:
            logging.error('Failed to restore model from \'{}\'.'.format(restoredir))
            exit(1)

    # ------------------------------------------------------------------ train loop (transformer.py:846-960)
    def train(self, dataset, input_shape, logdir, restoredir=None, epochs=None, learning_rate=1e-3,
              save_frequency_mode=ModelSaveFrequencyMode.EPOCH, save_frequency=1, max_checkpoints=1,
              show_progress_bar=True, max_steps=None, checkpoint_format='npz'):
        logdir = Path(logdir) if logdir is not None else None
        if restoredir is not None:
            logdir = Path(restoredir)                                            # :884-885
        rank = self._dp[0] if self._dp else 0
        manager = ckpt.CheckpointManager(logdir, max_to_keep=max_checkpoints, format=checkpoint_format)   # :890-891
        step, epoch = 1, 1
        if restoredir is not None:                                               # :894-900
            try:
                sd, meta = ckpt.load(manager.latest_checkpoint)
                self.load_state_dict(sd)
                step, epoch = int(meta['step']), int(meta['epoch'])
                logging.info('Model restored from \'{}\'.'.format(manager.latest_checkpoint))
            except Exception:
                logging.error('Failed to restore model from \'{}\'.'.format(restoredir))
                exit(1)
        summary = ckpt.ScalarLog(logdir / 'train') if rank == 0 else None        # :903
        save_frequency_mode = ModelSaveFrequencyMode(save_frequency_mode)
        history = []

        def save():
            if rank != 0:
                return None
            return manager.save(self.state_dict(), {'step': step, 'epoch': epoch})

        done = False
        while (epochs is None or epoch < epochs) and not done:                   # :907 (epoch starts at 1)
            logging.info('Epoch {}'.format(epoch if epochs is None else '{}/{}'.format(epoch, epochs)))
            ep_loss, ep_correct, ep_n, t0 = 0.0, 0.0, 0, time.time()
            # The device runs up to two steps ahead of this loop: step s is submitted (ids uploaded on the copy stream behind
            # step s-1) and the metrics of step s-1 are read and logged while s computes.  Logged values, their step numbers
            # and the checkpoint contents are those of the reference's synchronous loop (a save first drains the pipeline).
            pending = []                                                         # [(ticket, step number)]

            def retire(upto):
                nonlocal ep_loss, ep_correct
                while len(pending) > upto:
                    tk, st = pending.pop(0)
                    loss, acc = self.step_metrics(tk)
                    ep_loss += loss; ep_correct += acc
                    history.append((st, loss, acc))
                    if summary:
                        summary.scalar('loss', loss, st)                         # :933-936
                        summary.scalar('accuracy', acc, st)
                    if show_progress_bar and rank == 0 and (st % 10 == 1):
                        print('\r- loss: {:.4f} - accuracy: {:.4f}'.format(loss, acc), end='', flush=True)   # :939

            for x, y in dataset:                                                 # :914
                pending.append((self.train_step_async(x, y, learning_rate), step))
                ep_n += 1
                retire(1)
                if save_frequency_mode == ModelSaveFrequencyMode.GLOBAL_STEP and step % save_frequency == 0:
                    retire(0)
                    path = save()                                                # :941-943
                    if path and show_progress_bar:
                        print('\nSaved checkpoint for step {} at {}.'.format(step, path))
                step += 1                                                        # :945
                if max_steps is not None and len(history) + len(pending) >= max_steps:
                    done = True
                    break
            retire(0)
            if ep_n == 0:
                logging.error('The dataset yielded no batches.')
                break
            if summary:
                summary.scalar('epoch_loss', ep_loss / ep_n, epoch)              # :949-951
                summary.scalar('epoch_accuracy', ep_correct / ep_n, epoch)
            if save_frequency_mode == ModelSaveFrequencyMode.EPOCH and epoch % save_frequency == 0:
                path = save()                                                    # :953-955
                if path and show_progress_bar:
                    print('\nSaved checkpoint for epoch {} at {}.'.format(epoch, path))
            epoch += 1                                                           # :960
        if summary:
            summary.close()
        return history
