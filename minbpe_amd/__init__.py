"""minbpe_amd -- MI355X-native drop-in for karpathy/minbpe's train/encode path.

    from minbpe_amd import BasicTokenizer, RegexTokenizer

Using this package needs minbpe_amd/lib/libbpe_hip.so (python -m minbpe_amd.build)
and an AMD gfx950 GPU.  There is no CPU fallback: if the library is missing or stale
the first access to anything below raises ImportError.
"""
_EXPORTS = ("Tokenizer", "BasicTokenizer", "RegexTokenizer", "GPT4Tokenizer", "get_stats", "merge",
            "GPT2_SPLIT_PATTERN", "GPT4_SPLIT_PATTERN", "render_token", "replace_control_characters",
            "Engine", "synth_text", "split_offsets", "version")

try:
    from .tokenizer import (  # noqa: F401
        Tokenizer, BasicTokenizer, RegexTokenizer, GPT4Tokenizer, get_stats, merge,
        GPT2_SPLIT_PATTERN, GPT4_SPLIT_PATTERN, render_token, replace_control_characters,
    )
    from ._native import Engine, synth_text, split_offsets, version  # noqa: F401
    _load_error = None
except (ImportError, OSError, AttributeError) as _e:  # missing / stale .so: `python -m minbpe_amd.build`
    # (the package itself must stay importable so that its build module can run)
    _load_error = _e


def __getattr__(name):
    if _load_error is not None and name in _EXPORTS:
        raise ImportError(
            f"minbpe_amd: libbpe_hip.so could not be loaded ({_load_error}); "
            "build it with `python -m minbpe_amd.build` -- there is no CPU fallback") from _load_error
    raise AttributeError(f"module 'minbpe_amd' has no attribute {name!r}")
