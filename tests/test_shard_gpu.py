"""The N > 1 path with the real kernels: two ranks (gloo rendezvous on 127.0.0.1) share the box's one
GPU, each classifies its read block through the C ABI; counts are all-reduced, lists gathered."""
import pytest

from tests.test_shard_gloo import run_world

pytestmark = pytest.mark.gpu


def test_two_ranks_one_gpu():
    run_world(2, 300_001, 1, use_gpu=True)
    run_world(2, 70_000, 2, use_gpu=True)
