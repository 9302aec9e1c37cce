"""composer_amd: MI355X-native implementation of galacticglum/composer's Transformer hot path
(train step, evaluation, autoregressive decode) behind the reference's CLI / config / checkpoint surface."""
__version__ = '0.1.0'
