"""Importable alias of the `pets-face-recognition_amd/` package directory (a hyphen is not a valid module name).

`import pets_face_recognition_amd as pfr` resolves every submodule (`_hip`, `models`, `losses`, `engine`, `utils`,
`optim`, `match`, …) inside `pets-face-recognition_amd/`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pets-face-recognition_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
