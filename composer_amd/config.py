"""YAML configuration with dot access -- the surface of the reference's composer/config.py:8-72 (`Dotdict`,
`ConfigInstance`, `get`).  Same keys as the reference's default_config.yml; the optional `transformer.runtime`
block is new and defaults keep the reference meaning."""
import yaml


class Dotdict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __init__(self, data=None):
        for key, value in (data or {}).items():
            if hasattr(value, 'keys'):
                value = Dotdict(value)
            self[key] = value


class ConfigInstance(Dotdict):
    def __init__(self, filepath, data):
        self.filepath = filepath
        super().__init__(data)


def get(filepath):
    with open(filepath) as file:
        merged = {}
        for doc in yaml.safe_load_all(file):
            for k, v in (doc or {}).items():
                merged[k] = v
        return ConfigInstance(filepath, merged)
