"""Same import path as the reference's minbpe/base.py: the names live in tokenizer.py."""
from .tokenizer import Tokenizer, get_stats, merge, render_token, replace_control_characters  # noqa: F401
