"""Drop-in for the reference's Lexicon package (hanzi -> pinyin units -> pronunciation tree), SURVEY section 8(f) rank 3."""
from .PinYin import PinYin  # noqa: F401
from .PronunciationLexicon import PronunciationLexicon  # noqa: F401
