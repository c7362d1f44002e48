"""score_sde-style configs.  ``ml_collections`` is not installed in this image; ``ConfigDict`` below is the
attribute-dict subset of it that the reference's config files and models use."""


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
