"""Same import path as the reference's minbpe/gpt4.py (bpe / recover_merges: gpt4.py:11-46)."""
from .tokenizer import GPT4Tokenizer  # noqa: F401
from .tokenizer import _recover_merges as recover_merges  # noqa: F401
from .tokenizer import _split_by_ranks as bpe  # noqa: F401

GPT4_SPECIAL_TOKENS = GPT4Tokenizer.SPECIAL_TOKENS
