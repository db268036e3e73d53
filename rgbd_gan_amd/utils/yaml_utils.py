"""Config wrapper with the reference's contract (utils/yaml_utils.py:7-22): attribute access returns None for
missing keys -- many behaviours of the training loop are selected by ABSENT keys (config.rgb, config.lambda_rotate,
config.uniform_distribution, ...)."""
import yaml


class Config(object):
    def __init__(self, config_dict):
        object.__setattr__(self, "config", dict(config_dict))

    def __getattr__(self, key):
        cfg = object.__getattribute__(self, "config")
        return cfg[key] if key in cfg else None

    def __setattr__(self, key, value):
        if key == "config":
            object.__setattr__(self, key, value)
        else:
            self.config[key] = value

    def __getitem__(self, key):
        return self.config[key]

    def __repr__(self):
        return yaml.dump(self.config, default_flow_style=False)


def load(path):
    with open(path) as f:
        return Config(yaml.safe_load(f))
