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
