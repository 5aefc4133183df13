"""Import stub for gym (not installed here); see tests/golden/gen/make_golden.py."""
from . import utils  # noqa: F401
