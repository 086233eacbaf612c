"""Same import path as the reference's minbpe/basic.py."""
from .tokenizer import BasicTokenizer  # noqa: F401
