"""Same import path as the reference's minbpe/regex.py."""
from .tokenizer import GPT2_SPLIT_PATTERN, GPT4_SPLIT_PATTERN, RegexTokenizer  # noqa: F401
