"""Deterministic scripted 'LM' shared by tests/golden/make_golden.py and the parity tests.

After a context that is a prefix of `target` the arg-max is the next target token, otherwise a hash
of the last three tokens.  Every logits row is tie-free (a permutation of {0..V-1}/V plus a spike at
the arg-max and a smaller one at a runner-up), so top-k order is well defined.  V must be coprime
with 37.  numpy only.
"""
import numpy as np


class ScriptedLM:
    def __init__(self, target, vocab):
        assert np.gcd(37, vocab) == 1
        self.target, self.vocab = list(target), vocab

    def next_token(self, ctx):
        k = len(ctx)
        if k < len(self.target) and list(ctx) == self.target[:k]:
            return self.target[k]
        h = 1469598103
        for t in ctx[-3:]:
            h = (h * 1000003 + t + 7) % 2147483647
        return 3 + h % (self.vocab - 3)

    def row(self, nt):
        v = self.vocab
        r = ((np.arange(v) * 37 + 11 * nt) % v).astype(np.float32) / np.float32(v)
        r[nt] += np.float32(8.0)
        r[(nt * 7 + 1) % v] += np.float32(3.0)
        return r

    def logits(self, committed, tokens, anc):
        """[len(tokens), V] float32; node i sees committed + the tokens on its root->i path."""
        out = np.empty((len(tokens), self.vocab), np.float32)
        committed = list(committed)
        for i in range(len(tokens)):
            path, j = [], i
            while j != -1:
                path.append(tokens[j])
                j = anc[j]
            out[i] = self.row(self.next_token(committed + path[::-1]))
        return out


def perm_logits(rng, n, vocab):
    """tie-free random logits: every row a permutation of 0..V-1 (compact in JSON)."""
    return rng.permuted(np.tile(np.arange(vocab, dtype=np.float32), (n, 1)), axis=1)
