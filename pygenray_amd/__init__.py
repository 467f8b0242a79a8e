"""pygenray_amd -- MI355X-native drop-in for pygenray's ray-fan hot path."""
from . import _lib  # noqa: F401
