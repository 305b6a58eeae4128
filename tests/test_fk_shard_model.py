"""The sharded FK23 pipeline's index maps (which rank holds what, which twiddle, what the exchanges move) against the plain
three-transform pipeline of kzg::open_fk (reference src/kzg.rs:157-203), over Z_q instead of G1: CPU only, no GPU, no oracle."""
import random

import pytest

from fk_shard_model import Q, ShardModel, open_fk_plain, open_fk_split


@pytest.mark.parametrize("log2d,R", [(2, 2), (3, 2), (4, 2), (4, 4), (5, 4), (6, 8), (7, 8)])
def test_sharded_pipeline_equals_plain(log2d, R):
    rnd = random.Random(1000 * log2d + R)
    d = 1 << log2d
    srs = [rnd.randrange(Q) for _ in range(d)]
    m = ShardModel(srs, log2d, R)
    m.setup()
    for _ in range(2):                                  # hat_s is reused
        p = [rnd.randrange(Q) for _ in range(d)]
        assert m.open(p) == open_fk_plain(srs, p, log2d)


@pytest.mark.parametrize("log2d", [0, 1, 2, 3, 5, 7])
def test_split_pipeline_equals_plain(log2d):
    """the un-sharded device pipeline (even half without transforms, odd half through one inverse and one forward transform of size d)"""
    rnd = random.Random(77 + log2d)
    d = 1 << log2d
    srs = [rnd.randrange(Q) for _ in range(d)]
    p = [rnd.randrange(Q) for _ in range(d)]
    assert open_fk_split(srs, p, log2d) == open_fk_plain(srs, p, log2d)
