"""minbpe_amd -- MI355X-native drop-in for karpathy/minbpe's train/encode path.

    from minbpe_amd import BasicTokenizer, RegexTokenizer

Importing this package needs minbpe_amd/lib/libbpe_hip.so (python -m
minbpe_amd.build); using it needs an AMD gfx950 GPU.  There is no CPU fallback.
"""
from .tokenizer import (  # noqa: F401
    Tokenizer, BasicTokenizer, RegexTokenizer, GPT4Tokenizer, get_stats, merge,
    GPT2_SPLIT_PATTERN, GPT4_SPLIT_PATTERN, render_token, replace_control_characters,
)
from ._native import Engine, synth_text, version  # noqa: F401
