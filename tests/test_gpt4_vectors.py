"""GPU, needs the `tiktoken` package AND its cl100k_base rank file (neither is available offline:
the test skips, and parity with tiktoken stays "unpinned", SURVEY.md 8c / DESIGN.md 0(c)).

The two known-answer vectors the reference publishes for GPT4Tokenizer
(/root/reference/README.md:40-72; its tests/test_tokenizer.py:62-77 compares with tiktoken live)."""
import pytest

pytestmark = pytest.mark.gpu

README_TEXT = "hello123!!!? (안녕하세요!) 😉"
README_IDS = [15339, 4513, 12340, 30, 320, 31495, 230, 75265, 243, 92245, 16715, 57037]
SPECIAL_TEXT = "<|endoftext|>hello world"
SPECIAL_IDS = [100257, 15339, 1917]


@pytest.fixture(scope="module")
def gpt4():
    tiktoken = pytest.importorskip("tiktoken")
    try:
        tiktoken.get_encoding("cl100k_base")
    except Exception as e:  # no network, no cached rank file
        pytest.skip(f"cl100k_base ranks unavailable: {e}")
    from minbpe_amd import GPT4Tokenizer
    return GPT4Tokenizer()


def test_readme_vector(gpt4):
    assert gpt4.encode(README_TEXT) == README_IDS
    assert gpt4.decode(README_IDS) == README_TEXT


def test_readme_special_tokens_vector(gpt4):
    assert gpt4.encode(SPECIAL_TEXT, allowed_special="all") == SPECIAL_IDS
    assert gpt4.decode(SPECIAL_IDS) == SPECIAL_TEXT
