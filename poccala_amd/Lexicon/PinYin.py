"""Drop-in for Lexicon/PinYin.py: hanzi -> pinyin in the reference's initial/final unit spelling.

Class surface mirrored: PinYin(path) (Lexicon/PinYin.py:20-37), word2pinyin(string, separate, check_tone, extend,
show_tone_mark) (:58-80).  The table is the reference's own data format: `Mandarin.dat`, one line per code point,
`HEX<TAB>PIN1 PIN2 ...` with tone digits 1-5 (:39-56); the file is data and is read from wherever the caller keeps
it (POCCALA_MANDARIN_DAT or the `path` argument) -- it is not shipped here.

The rewriting rules are restated from what `__check_tone` (:82-132) does to each reading, quirks included, and are
pinned by golden vectors produced by running the reference class (tests/golden/make_golden_lexicon.py):
  separate    'zh|ch|sh' or a one-letter initial (y and w count) is split off with a comma: 'zhong1' -> 'zh,ong1'
  check_tone  after j/q/x every 'u' becomes 'v' unless the reading contains 'iu'; 'ue' always becomes 've'
  extend      'y' -> '#_I', 'w' -> '#_u'; otherwise tone 5 is rewritten to 0 and a zero-initial reading gets its
              '#_a' / '#_o' / '#_e' / '#_v' initial -- looked up WITH the tone digit when tones are shown, so with
              show_tone_mark=True it never matches (reference behaviour: 'an1' stays 'an1')
  no extend   a reading containing y or w loses its first character
  show_tone_mark=False   tone digits are dropped at the end and duplicates merged (order unspecified, as in the reference)
"""
import os

_INITIALS = ('b', 'p', 'm', 'f', 'd', 't', 'n', 'l', 'g', 'k', 'h', 'j', 'q', 'x', 'zh', 'ch', 'sh', 'z', 'c', 's', 'r', 'y', 'w')
_ZERO_INITIAL = {'ai': '#_a', 'ao': '#_a', 'an': '#_a', 'ang': '#_a', 'o': '#_o', 'ou': '#_o', 'e': '#_e', 'ei': '#_e',
                 'er': '#_e', 'en': '#_e', '?': '#_v'}


def default_table_path():
    return os.environ.get('POCCALA_MANDARIN_DAT', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'Mandarin.dat'))


class PinYin(object):
    def __init__(self, path=None):
        self.__table = {}
        with open(path or default_table_path(), 'r') as f:
            for line in f:
                fields = line.strip('\n').split('\t')
                if len(fields) < 2:
                    continue
                self.__table[fields[0].lower()] = [r.lower() for r in fields[1].split(' ')]

    def readings(self, ch):
        """Raw readings of one character ('jia1', ...), or None when the table does not hold it."""
        return self.__table.get('%x' % ord(ch))

    @staticmethod
    def _rewrite(reading, separate, check_tone, extend, show_tone_mark):
        s = reading
        if separate and s[0] in _INITIALS:
            cut = 2 if (len(s) >= 3 and s[:2] in _INITIALS) else 1
            s = s[:cut] + ',' + s[cut:]
        if check_tone:
            if s[0] in ('j', 'q', 'x') and 'u' in s and 'iu' not in s:
                s = s.replace('u', 'v')
            if 'ue' in s:
                s = s.replace('ue', 've')
        if extend:
            if 'y' in s:
                s = s.replace('y', '#_I')
            elif 'w' in s:
                s = s.replace('w', '#_u')
            else:
                if show_tone_mark:
                    if int(s[-1]) == 5:
                        s = s[:-1] + '0'
                    key = s
                else:
                    key = s[:-1]
                head = _ZERO_INITIAL.get(key)
                if head is not None:
                    s = head + (',' if separate else '') + s
        elif 'y' in s or 'w' in s:
            s = s[1:]
        return s

    def word2pinyin(self, string, separate=True, check_tone=True, extend=True, show_tone_mark=True):
        """One list of unit spellings per character, every reading of a polyphonic character kept; None if a character
        is not in the table (the reference catches the KeyError, PinYin.py:79-80)."""
        out = []
        for ch in string:
            raw = self.readings(ch)
            if raw is None:
                return None
            got = [self._rewrite(r, separate, check_tone, extend, show_tone_mark) for r in raw]
            if not show_tone_mark:
                got = list(set(g[:-1] for g in got))
            out.append(got)
        return out
