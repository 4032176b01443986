"""`lib` of the drop-in layout (the reference's lib/ is a namespace package without __init__.py).

Imported as `uda_poseestimation_amd.lib` this file does nothing.  Imported as the TOP-LEVEL `lib` (this package's directory
is first on sys.path, the way INTEGRATION.md binds train_human.py:19-28), it registers the package's modules under the
reference's names (`lib`, `lib.models`, `lib.models.loss`, `lib.models.Style_net`, `lib.keypoint_detection`, `utils`) and
replaces itself in sys.modules, so both spellings name the same module objects."""
if __name__ == "lib":
    import _dropin
    _dropin.alias()
