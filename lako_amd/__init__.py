"""lako_amd — MI355X-native Fusion-in-Decoder reader behind LaKo's FiDT5 / train_reader.py API."""
from .config import FiDConfig  # noqa: F401
from .model import FiDT5  # noqa: F401
from .retriever import Retriever, RetrieverConfig  # noqa: F401

__all__ = ["FiDConfig", "FiDT5", "Retriever", "RetrieverConfig"]
