"""`ImageVisual`: inert stand-in (see the package docstring) — constructing or calling it shows nothing."""


class ImageVisual:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *images, **kwargs):
        pass
