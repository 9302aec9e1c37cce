"""YAML configuration with attribute access.

Compatibility surface of the reference's composer/config.py:8-72: the names `Dotdict`, `ConfigInstance`
(with `.filepath`) and `get(filepath)`, and the behaviour the CLI relies on -- `config.transformer.model.window_size`
as well as `config['transformer']['model']`, `.get(key, default)`, and assignment through either notation.  The
implementation is this package's own: nested mappings are wrapped on the way in (`_wrap`), a missing key read as an
attribute raises AttributeError (so `hasattr`, `copy.copy` and `pickle` behave; the reference's raises KeyError),
and names that start with an underscore never become keys.  Same keys as the reference's default_config.yml; the
optional `transformer.runtime` block is new and defaults keep the reference meaning.
"""
from collections.abc import Mapping

import yaml


def _wrap(value):
    """Mappings (at any depth, also inside lists) become Dotdicts; everything else is kept as is."""
    if isinstance(value, Dotdict):
        return value
    if isinstance(value, Mapping):
        return Dotdict(value)
    if isinstance(value, list):
        return [_wrap(v) for v in value]
    return value


class Dotdict(dict):
    """A dict whose string keys can also be read, written and deleted as attributes."""

    def __init__(self, data=None, **more):
        super().__init__()
        self.update(data or {}, **more)

    # every way into the dict goes through _wrap, so nested access works however the value arrived
    def __setitem__(self, key, value):
        super().__setitem__(key, _wrap(value))

    def update(self, *args, **kwargs):
        for key, value in dict(*args, **kwargs).items():
            self[key] = value

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def __getattr__(self, name):
        # only called when normal attribute lookup failed
        if name.startswith('_') or name not in self:
            raise AttributeError('%s has no key %r' % (type(self).__name__, name))
        return self[name]

    def __setattr__(self, name, value):
        if name.startswith('_'):
            object.__setattr__(self, name, value)
        else:
            self[name] = value

    def __delattr__(self, name):
        if name.startswith('_') or name not in self:
            raise AttributeError(name)
        del self[name]

    def __dir__(self):
        return sorted(set(super().__dir__()) | {k for k in self if isinstance(k, str)})

    def to_dict(self):
        """Plain nested dicts again (what yaml.safe_dump wants)."""
        def plain(v):
            if isinstance(v, Mapping):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, list):
                return [plain(x) for x in v]
            return v
        return plain(self)


class ConfigInstance(Dotdict):
    """A loaded configuration file; `filepath` is where it came from (`cli.train` copies that file next to the
    checkpoints, reference cli.py:552-577) and is not one of the configuration's keys."""

    def __init__(self, filepath, data):
        object.__setattr__(self, '_filepath', filepath)
        super().__init__(data)

    @property
    def filepath(self):
        return self._filepath

    def __reduce__(self):
        return (ConfigInstance, (self._filepath, self.to_dict()))


def get(filepath):
    """Loads every YAML document of `filepath`; later documents override earlier top-level keys."""
    merged = {}
    with open(filepath) as stream:
        for document in yaml.safe_load_all(stream):
            if document:
                merged.update(document)
    return ConfigInstance(filepath, merged)
