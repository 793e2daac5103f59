"""Configuration surface of the reference (config.py:1-31): same names, same defaults."""

# jobs that may run at once PER GPU (the reference runs everything on device 0; here the
# scheduler multiplies this by the number of GPUs of the node). Use 1 when levels_num > 2.
simultaneous_tasks_count = 2

_DEFAULTS = dict(
    content_weight=1e3,            # weight of the content loss
    style_weight=4e5,              # weight of the style loss
    tv_weight=1e2,                 # weight of the total-variation loss
    optimizer="lbfgs",             # 'lbfgs' | 'adam'
    model="vgg19",                 # 'vgg19'
    init_method="content+noise",   # 'random' | 'content+noise' | 'style'
    levels_num=2,                  # pyramid levels (4 for maximum resolution)
    iters_num=500,                 # closure evaluations (1500 for maximum quality)
    noise_factor=0.95,             # strength of the noise blended into the initial image
    noise_levels=(9, 18, 36, -1, 0),                               # spots along the short axis / spot size / 0 = constant
    noise_levels_central_amplitude=(0.30, 0.20, 0.10, 0.20, 0.20),
    noise_levels_peripheral_amplitude=(0.20, 0.30, 0.40, 0.10, 0.00),
    noise_levels_dispersion=(0.20, 0.30, 0.40, 0.60, 0.30),
)


class Config:
    """Settings of one style-transfer job; positional or keyword arguments in the reference's order
    (config.py:5-18): Config(1e3, 4e5, 1e2, 'adam') and Config(optimizer='adam') both work."""

    def __init__(self, *args, **kwargs):
        names = list(_DEFAULTS)
        if len(args) > len(names):
            raise TypeError(f"Config() takes at most {len(names)} positional arguments ({len(args)} given)")
        for name, value in zip(names, args):
            if name in kwargs:
                raise TypeError(f"Config() got multiple values for argument '{name}'")
            kwargs[name] = value
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"Config() got unexpected keyword argument(s): {sorted(unknown)}")
        for name, default in _DEFAULTS.items():
            setattr(self, name, kwargs.get(name, default))

    def __repr__(self):
        return "Config(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _DEFAULTS) + ")"
