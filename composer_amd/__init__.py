"""composer_amd: MI355X-native implementation of galacticglum/composer's Transformer hot path
(train step, evaluation, autoregressive decode) behind the reference's CLI / config / checkpoint surface."""
import os as _os

# multi-process GPU work on this pool (RCCL ranks) needs dmabuf IPC; must be in the environment before HIP initialises
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = '0.1.0'
