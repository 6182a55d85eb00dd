"""The scheduler of the lock-step lanes (csrc/batch.cpp) without a device: provers of a batch run as fibers of one thread; a fiber
parks when it waits for the stream or votes, and the per-thread error string behind zkhip_last_error() must stay each fiber's own."""
import pytest

from zktls_amd import _lib


@pytest.mark.parametrize("members,rounds", [(1, 1), (2, 8), (3, 5), (16, 12), (64, 40), (256, 9)])
def test_fibers_wait_vote_and_leave(members, rounds):
    assert _lib.load().zkhip_selftest_lockstep(members, rounds) == 0


def test_selftest_refuses_nonsense():
    L = _lib.load()
    assert L.zkhip_selftest_lockstep(0, 4) == 1 and L.zkhip_selftest_lockstep(257, 4) == 1 and L.zkhip_selftest_lockstep(4, 0) == 1


def test_every_kernel_launch_of_the_prover_goes_through_the_batcher():
    """a kernel launched with hipLaunchKernelGGL from a source of the proving path would run unmerged AND unordered with respect to the
    parked members' requests only in program order -- allowed for table builders, not for the prover's kernels: the list of direct
    launches is pinned here, so a new one is a decision"""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zktls_amd", "csrc")
    direct = {}
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".cpp")):
            names = re.findall(r"hipLaunchKernelGGL\(\(?([\w:]+)", open(os.path.join(root, f)).read())
            if names:
                direct[f] = sorted(set(names))
    direct.pop("hal.hip", None)                                   # the RISC Zero operator surface: not on the lock-step path
    direct.pop("ntt_fused.hip", None)                             # 2^20-row shapes only
    assert direct == {"hash.hip": ["mrec_chains16_kernel", "mrec_chains_kernel"],        # round 6: the recursion machines' witness kernels (with the ones in the .inl files) -- never launched inside a batch:
                                                                  # shard_verifier_prove_impl / top_begin take the host's walk when t_batcher is set
                      "ntt.hip": ["combine_table_kernel", "ntt_colpass_kernel", "ntt_pass1024x2_kernel", "post2d_table_kernel",
                                  "post_table_kernel", "pow_table_kernel"]}      # plan tables, the column-major pass, an A/B kernel
