"""Scene-group streams are chosen by measurement (mujoco_rl_ur5_amd/streams.py): the HIP runtime decides at a stream's first use which of its four hardware queues it
gets, and two group streams on one queue run the groups' launches strictly one after the other (profiles/r05_q_bench_queue_map.txt)."""
import pytest
import torch

from mujoco_rl_ur5_amd import streams


def test_without_a_gpu_nothing_is_claimed():
    s, verified = streams.group_streams(torch, torch.device("cpu"), 2)
    assert s == [None, None] and verified is False


@pytest.mark.gpu
def test_group_streams_overlap_on_the_gpu():
    dev = torch.device("cuda", 0)
    junk = [torch.cuda.Stream(device=dev) for _ in range(5)]                  # an odd stream history before the groups are made
    for j in junk:
        with torch.cuda.stream(j):
            torch.zeros(8, device=dev)
    s, verified = streams.group_streams(torch, dev, 2)
    assert verified and len(s) == 2 and s[0].cuda_stream != s[1].cuda_stream
    cycles = 1 << 17
    alone = streams._spin_ms(torch, s[:1], cycles)
    while alone < 2.0:
        cycles *= 4
        alone = streams._spin_ms(torch, s[:1], cycles)
    both = min(streams._spin_ms(torch, s, cycles) for _ in range(3))
    assert both < 1.5 * alone, (alone, both)                                 # two spin kernels side by side, not one after the other
    one, ok = streams.group_streams(torch, dev, 1)
    assert ok and len(one) == 1
