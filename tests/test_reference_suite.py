"""CPU, build container only: the reference's OWN test module (tests/test_tokenizer.py, read from
its read-only tree at run time, never copied) executed with `minbpe` resolving to this package.
The device engine is the oracle-backed test double (tests/fake_engine.py), so this exercises the
drop-in surface -- names, signatures, results, exceptions, save/load -- not the kernels; the two
tiktoken-equality tests are left out (no cl100k ranks offline)."""
import importlib.util
import os
import sys
import types

import pytest

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_test_module_passes_against_this_package(native, monkeypatch, tmp_path):
    import minbpe_amd
    import minbpe_amd.tokenizer as T
    from fake_engine import OracleEngine
    eng = OracleEngine()
    monkeypatch.setattr(T, "engine", lambda device=None: eng)
    monkeypatch.setitem(sys.modules, "minbpe", minbpe_amd)  # `from minbpe import BasicTokenizer, ...`
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    spec = importlib.util.spec_from_file_location("ref_test_tokenizer", os.path.join(REF, "tests", "test_tokenizer.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    assert ref.BasicTokenizer is minbpe_amd.BasicTokenizer and ref.RegexTokenizer is minbpe_amd.RegexTokenizer
    ran = 0
    for factory in (ref.BasicTokenizer, ref.RegexTokenizer):
        for text in ref.test_strings:  # includes FILE:taylorswift.txt, resolved next to the reference's test file
            ref.test_encode_decode_identity(factory, text)
            ran += 1
        ref.test_wikipedia_example(factory)
        ran += 1
    monkeypatch.chdir(tmp_path)  # test_save_load writes its files into the working directory
    for specials in ({}, ref.special_tokens):
        ref.test_save_load(specials)
        ran += 1
    assert ran == 12 and os.listdir(tmp_path) == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_saved_files_are_byte_identical_to_the_reference(native, monkeypatch, tmp_path):
    """train.py's flow (train, register specials, save) on a slice of the reference's own sample
    text: the .model and .vocab files written by this package and by the reference are the same bytes."""
    import minbpe_amd.tokenizer as T
    from fake_engine import OracleEngine
    eng = OracleEngine()
    monkeypatch.setattr(T, "engine", lambda device=None: eng)
    monkeypatch.setitem(sys.modules, "tiktoken", types.ModuleType("tiktoken"))
    monkeypatch.syspath_prepend(REF)
    import minbpe as ref
    text = open(os.path.join(REF, "tests", "taylorswift.txt"), encoding="utf-8").read()[:40_000]
    text += "\x00\x07 control ​ chars � and a lone \x80 byte".encode("utf-8", "surrogatepass").decode("utf-8", "replace")
    for name, ours, theirs in (("basic", T.BasicTokenizer, ref.BasicTokenizer),
                               ("regex", T.RegexTokenizer, ref.RegexTokenizer)):
        a, b = ours(), theirs()
        a.train(text, 256 + 80)
        b.train(text, 256 + 80)
        assert a.merges == b.merges and a.vocab == b.vocab
        if name == "regex":
            a.register_special_tokens({"<|endoftext|>": 100257, "<|pad|>": 100258})
            b.register_special_tokens({"<|endoftext|>": 100257, "<|pad|>": 100258})
        a.save(str(tmp_path / f"ours_{name}"))
        b.save(str(tmp_path / f"ref_{name}"))
        for ext in (".model", ".vocab"):
            assert open(tmp_path / f"ours_{name}{ext}", "rb").read() == open(tmp_path / f"ref_{name}{ext}", "rb").read()
        # and each side loads the other's file
        c = ours()
        c.load(str(tmp_path / f"ref_{name}.model"))
        assert c.merges == b.merges and c.special_tokens == b.special_tokens and c.pattern == b.pattern
        probe = text[1000:3000]
        assert c.encode(probe) == b.encode(probe) and c.decode(c.encode(probe)) == probe
