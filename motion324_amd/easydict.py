"""Minimal dict-with-attribute-access, the return type contract of Motion_Latent_Model.forward.

The reference returns ``easydict.EasyDict`` (model/Pcd_motion.py:584-597); callers rely on
``isinstance(out, dict)``, ``'pcd_moved' in out``, ``out.loss_metrics.loss`` and ``.items()``
(scripts/inference_with_video_mesh.py:169, train.py:162,171,223).  ``easydict`` itself is not
installed in this image, so the package carries its own 20-line equivalent.
"""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, EasyDict):
            value = EasyDict(value)
        super().__setitem__(key, value)

    __setattr__ = __setitem__

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError:
            raise AttributeError(key) from None
