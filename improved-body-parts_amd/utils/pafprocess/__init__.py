"""Drop-in for the reference package `utils.pafprocess` (SWIG module `pafprocess`): the import line
`from utils.pafprocess import pafprocess` (evaluate.py:8, demo_image.py:24) keeps working."""
from . import pafprocess  # noqa: F401
