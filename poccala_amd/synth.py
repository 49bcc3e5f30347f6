"""Synthetic corpus / model generators for the BASELINE.json configurations.

Seeds: features 0, model 1, labels 2 (np.random.default_rng).  Features x ~ N(0,1) float32,
mu ~ N(0,1), var ~ U(0.5,2), weights ~ Dirichlet(1), unit transitions = the reference's flat start
(AcousticModel/AcousticModel.py:174-181).  Used by bench.py and the tests; no speech data ships.
"""
import numpy as np

S = 5  # state_num: entry + 3 emitting + exit (AcousticModel.py:39)

# BASELINE.json configs -> canonical shapes (SURVEY.md section 8, table at the top)
CONFIGS = {
    'C1': dict(U=1, T=300, D=13, M=4, units=1, L=1),
    'C2': dict(U=128, T=300, D=39, M=256, units=50, L=20),
    'C3': dict(U=1024, T=300, D=39, M=2048, units=50, L=20),
    'C4shard': dict(U=1024, T=300, D=39, M=2048, units=1000, L=20),   # C4 = 8 such shards, one per GPU
    'C5shard': dict(U=417, T=300, D=39, M=4096, units=183, L=20),
}


def flat_start_transmat(s=S):
    a = np.zeros((s, s))
    a[0, 1] = 1.0
    for j in range(1, s - 1):
        a[j, j] = 0.5
        a[j, j + 1] = 0.5
    return a


def random_left_right_transmat(rng, s=S):
    """a left-to-right unit matrix with its own self-loop probabilities (what a transition M-step leaves): tests use it where every unit
    having the SAME flat-start matrix would hide a mix-up between utterances or units."""
    a = np.zeros((s, s))
    a[0, 1] = 1.0
    for j in range(1, s - 1):
        x = rng.uniform(0.2, 0.8)
        a[j, j] = x
        a[j, j + 1] = 1.0 - x
    return a


def make_model(units, M, D, seed=1, s=S, dtype=np.float64):
    """(mean, var, weight) for J = units*(S-2) GMM states, and the per-unit transition matrices."""
    rng = np.random.default_rng(seed)
    J = units * (s - 2)
    mean = rng.standard_normal((J, M, D), dtype=np.float32).astype(dtype)
    var = rng.uniform(0.5, 2.0, (J, M, D)).astype(dtype)
    w = rng.standard_exponential((J, M))
    w /= w.sum(axis=1, keepdims=True)          # Dirichlet(1)
    trans = [flat_start_transmat(s) for _ in range(units)]
    return mean, var, w, trans


def make_frames(U, T, D, seed=0, ragged=False):
    """Concatenated (F,D) float32 frames, per-utterance lengths and start rows."""
    rng = np.random.default_rng(seed)
    if ragged:
        lens = rng.integers(max(2, (2 * T) // 3), (4 * T) // 3 + 1, size=U).astype(np.int32)
    else:
        lens = np.full(U, T, dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    frames = rng.standard_normal((int(lens.sum()), D), dtype=np.float32)
    return frames, lens, begin


def make_peaked_frames(labels, T, mean, var, seed=5, s=S):
    """Features sampled from the model ALONG each utterance's label (one Gaussian of the state the frame is aligned to):
    the peaked posteriors of aligned speech, where a tenth of the (frame, state) pairs survive the E-step's underflow cut."""
    rng = np.random.default_rng(seed)
    e = s - 2
    M, D = mean.shape[1], mean.shape[2]
    fr = np.empty((len(labels) * T, D), dtype=np.float32)
    for u, lab in enumerate(labels):
        per = max(1, T // (e * len(lab)))
        st = np.repeat(np.asarray(lab)[:, None] * e + np.arange(e)[None, :], per).reshape(-1)[:T]
        st = np.concatenate([st, np.full(T - len(st), st[-1])])
        mix = rng.integers(0, M, size=T)
        fr[u * T:(u + 1) * T] = mean[st, mix] + np.sqrt(var[st, mix]) * rng.standard_normal((T, D))
    return fr


def make_labels(U, L, units, seed=2):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, units, size=L) for _ in range(U)]


def make_pronunciation_tree(n_words, units_n, seed=55, fixture=None):
    """A synthetic pronunciation tree for BASELINE config 5 (the reference ships no word list): the words of the golden
    lexicon fixture plus n_words random strings of 1-4 of its characters, read through the reference's Mandarin.dat rules
    (poccala_amd.Lexicon), compiled against units_n unit ids.  Returns (tree, lexicon)."""
    import json
    import os
    import tempfile
    from .Lexicon import PinYin, PronunciationLexicon
    if fixture is None:
        fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'G13_lexicon.json')
    g = json.load(open(fixture))
    with tempfile.NamedTemporaryFile('w', suffix='.dat', delete=False) as f:
        for k, v in g['table'].items():
            f.write('%s\t%s\n' % (k, v))
        tab = f.name
    try:
        py = PinYin(tab)
    finally:
        os.unlink(tab)
    chars = sorted({ch for w in g['words'] for ch in w})
    rng = np.random.default_rng(seed)
    words = list(g['words']) + [''.join(rng.choice(chars, size=rng.integers(1, 5))) for _ in range(n_words)]
    lx = PronunciationLexicon()
    lx.generate_lexicon(words=words, pinyin=py)
    names = sorted({u for w in words for r in (py.word2pinyin(w) or []) for x in r for u in x.split(',')})
    names = names[:units_n] + ['pad%d' % i for i in range(max(0, units_n - len(names)))]
    return lx.compile({u: i for i, u in enumerate(names)}), lx
