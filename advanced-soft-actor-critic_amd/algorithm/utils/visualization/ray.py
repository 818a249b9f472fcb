"""`RayVisual`: inert stand-in (see the package docstring) — constructing or calling it shows nothing."""


class RayVisual:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *rays, **kwargs):
        pass
