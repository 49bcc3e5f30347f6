"""Drop-in for Lexicon/PronunciationLexicon.py: the pronunciation tree the decoder walks, and its flat form for the GPU.

Mirrored (Lexicon/PronunciationLexicon.py:25-94): `generate_lexicon` builds a nested dict
    lexicon[initial of the first character][reading of character 1][reading of character 2]...['word'] = [words]
with one branch per combination of readings of polyphonic characters (__create_tree :79-94); `init_lexicon` loads a
pickled tree (:30-39; the reference's `lexicon_ch_small.pkl` and word lists are not shipped, any word list works);
`lexicon` returns the dict.  Pinned by golden vectors made by running the reference class on a word list
(tests/golden/make_golden_lexicon.py).

`compile` flattens the tree for pcl_lexicon_upload: one node per reading below the initial level, its units (the
reading split at the comma: initial + final, or a lone final), its children and whether words end there.
"""
import os
import pickle

import numpy as np

from .PinYin import PinYin


class PronunciationLexicon(object):
    def __init__(self):
        self.__lexicon = {}
        self.__lexicon_size = 0
        self.skipped = []        # words with a character outside the table (the reference crashes on them)

    @property
    def lexicon(self):
        return self.__lexicon

    @property
    def size(self):
        return self.__lexicon_size

    def init_lexicon(self, loadpath):
        with open(loadpath, 'rb') as f:
            self.__lexicon = pickle.load(f)

    def save_lexicon(self, savepath):
        with open(savepath, 'wb') as f:
            pickle.dump(self.__lexicon, f)

    def generate_lexicon(self, path=None, savepath=None, words=None, pinyin=None):
        """path: directory of word-list files, one word per line, read until the first empty line of each file
        (PronunciationLexicon.py:54-58,73); or `words`: an iterable of words."""
        pinyin = pinyin or PinYin()
        if words is None:
            words = []
            for root, _, files in os.walk(path):
                for name in files:
                    with open(os.path.join(path, name), 'r') as f:
                        for line in f:
                            w = line.strip('\n')
                            if not w:
                                break
                            words.append(w)
        for w in words:
            p = pinyin.word2pinyin(w, separate=True, check_tone=True, extend=True, show_tone_mark=True)
            if p is None:
                self.skipped.append(w)
                continue
            self.__lexicon_size += 1
            for reading in p[0]:
                level1 = self.__lexicon.setdefault(reading.split(',')[0], {})
                self._grow(level1.setdefault(reading, {}), p[1:], w)
        if savepath:
            self.save_lexicon(savepath)
        return self.__lexicon

    @staticmethod
    def _grow(node, rest, word):
        if not rest:
            ws = node.setdefault('word', [])
            if word not in ws:
                ws.append(word)
            return
        for reading in rest[0]:
            PronunciationLexicon._grow(node.setdefault(reading, {}), rest[1:], word)

    # ------------------------------------------------------------------ flat form for the device
    def compile(self, unit_index, max_units=2):
        """unit_index: {unit name: id} (the acoustic model's inventory, AcousticModel.load_unit).  Returns a dict of
        arrays: node_units (n, max_units) int32 (-1 padded), node_nunits, node_parent (-1 for first characters),
        child_ptr (n+1) / child_idx (children in insertion order, as dict iteration gives them to the reference),
        node_word (1 where words end), roots (first-character nodes), `names` / `words` for reporting.  Branches with a
        unit outside the inventory are left out (`dropped`)."""
        units, parent, names, words, kids = [], [], [], [], []
        dropped = []

        def add(reading, node, par):
            label = reading.split(',')
            if len(label) > max_units or any(u not in unit_index for u in label):
                dropped.append(reading)
                return -1
            me = len(units)
            units.append([unit_index[u] for u in label] + [-1] * (max_units - len(label)))
            parent.append(par)
            names.append(reading)
            words.append(list(node.get('word', [])))
            kids.append([])
            for key, sub in node.items():
                if key == 'word':
                    continue
                c = add(key, sub, me)
                if c >= 0:
                    kids[me].append(c)
            return me
        roots = []
        for initial, level1 in self.__lexicon.items():
            for reading, node in level1.items():
                r = add(reading, node, -1)
                if r >= 0:
                    roots.append(r)
        n = len(units)
        child_ptr = np.zeros(n + 1, dtype=np.int32)
        for i in range(n):
            child_ptr[i + 1] = child_ptr[i] + len(kids[i])
        child_idx = np.array([c for k in kids for c in k], dtype=np.int32)
        nu = np.array([sum(1 for u in row if u >= 0) for row in units], dtype=np.int32) if n else np.zeros(0, np.int32)
        return dict(node_units=np.array(units, dtype=np.int32).reshape(n, max_units), node_nunits=nu,
                    node_parent=np.array(parent, dtype=np.int32), child_ptr=child_ptr, child_idx=child_idx,
                    node_word=np.array([1 if w else 0 for w in words], dtype=np.int32), roots=np.array(roots, dtype=np.int32),
                    names=names, words=words, dropped=dropped)
