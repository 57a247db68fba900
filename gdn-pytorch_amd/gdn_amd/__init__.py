"""gdn_amd -- MI355X-native hot path of GDN-Pytorch (host-side mirror of the reference interface)."""
from ._lib import GdnError, LIB_PATH  # noqa: F401

__all__ = ["GdnError", "LIB_PATH"]
