import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


class ConfigMixin:
    """register_to_config stores ctor kwargs in self.config; unknown attributes fall through to it."""

    config_name = "config.json"

    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_internal_dict", {}))
        cfg.update(kwargs)
        object.__setattr__(self, "_internal_dict", FrozenDict(cfg))

    @property
    def config(self):
        return self._internal_dict

    def __getattr__(self, name):
        # nn.Module.__getattr__ first (parameters / buffers / submodules), then config fall-through
        try:
            return super().__getattr__(name)
        except AttributeError:
            d = self.__dict__.get("_internal_dict")
            if d is not None and name in d:
                return d[name]
            raise

    @classmethod
    def from_config(cls, config, **kwargs):
        config = dict(config)
        config.update(kwargs)
        sig = inspect.signature(cls.__init__).parameters
        init_kwargs = {k: v for k, v in config.items() if k in sig and k != "self"}
        hidden = {k: v for k, v in config.items() if k not in sig and not k.startswith("_")}
        model = cls(**init_kwargs)
        model.register_to_config(**hidden)
        return model

    @classmethod
    def load_config(cls, path, **kwargs):
        import json
        with open(path) as f:
            return json.load(f)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update(kwargs)
        init(self, *args, **kwargs)
        self.register_to_config(**cfg)

    return inner
