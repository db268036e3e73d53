"""Configuration object with the reference's contract (utils/yaml_utils.py:7-22): a YAML mapping whose keys read as
attributes, and a MISSING key reads as None -- many behaviours of the training loop are selected by absent keys
(config.rgb, config.lambda_rotate, config.uniform_distribution, config.keep_smoothed_gen, ...)."""
import yaml


class Config(dict):
    """dict with attribute access; `cfg.some_key` is `cfg.get("some_key")`, so unknown keys are None, not errors."""

    def __getattr__(self, key):
        if key.startswith("__"):           # keep copy / pickle protocol probes honest
            raise AttributeError(key)
        return self.get(key)

    def __setattr__(self, key, value):
        self[key] = value

    @property
    def config(self):                      # the reference exposes the underlying mapping under this name
        return self

    def __repr__(self):
        return yaml.dump(dict(self), default_flow_style=False)


def load(path):
    """Read a YAML file into a Config."""
    with open(path) as stream:
        return Config(yaml.safe_load(stream) or {})
