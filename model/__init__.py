"""Drop-in import path of the reference (``from model.SGDiff import SGDiff`` in scripts/eval_3dfront.py:16).
The implementation lives in echoscene_amd/model/scene.py."""
