"""HIP streams for scene groups that really run side by side.

Scene groups (bench.Job, BatchedGraspAgent(pipeline_groups=G)) hide one group's launch tail and host-side gaps under the other group's kernels -- which only happens
when the groups' streams sit on DIFFERENT hardware queues. The HIP runtime deals its (four) hardware queues to streams when a stream is first used, by the load the
queues carry at that moment: which queue a stream gets depends on every stream the process created, used and destroyed before. Measured twice in round 5
(profiles/r05_f_dqn512_timeline_one_queue.txt, profiles/r05_q_bench_queue_map.txt): both group streams of a job landed on one queue and the groups' launches ran
strictly one after the other (six-object rounds: 1 109 ms instead of 712 ms), after a change that touched no stream at all. So the streams are CHOSEN by measurement:
a stream is accepted into the set if a spin kernel on it overlaps spin kernels on the streams already chosen."""
import time


def _spin_ms(torch, streams, cycles):
    """Wall time of one spin kernel per stream, all queued before any is waited for."""
    for s in streams:
        s.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(cycles))
    for s in streams:
        s.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def group_streams(torch, device, n, tries=16, first_high_priority=False):
    """``n`` torch streams on ``device`` whose kernels overlap, and whether that was verified: (streams, verified). One stream: torch's current one. Without a GPU
    (or without torch's spin kernel) the streams are returned unverified. first_high_priority: stream 0 is a high-priority stream -- when two groups' launches are BOTH
    larger than the chip (40-object piles: 2048 scenes per group, 512 resident), equal priorities make the dispatcher alternate between the two queues, the two launches
    advance at one rate and their tails coincide; with group 0 ahead, each group's tail is filled by the other's bulk."""
    if device.type != "cuda":
        return [None] * n, False
    if n <= 1:
        return [torch.cuda.current_stream(device)], True
    with torch.cuda.device(device):
        chosen = [torch.cuda.Stream(device=device, priority=-1) if first_high_priority else torch.cuda.Stream(device=device)]
        if not hasattr(torch.cuda, "_sleep"):
            return chosen + [torch.cuda.Stream(device=device) for _ in range(n - 1)], False
        cycles = 1 << 17
        _spin_ms(torch, chosen, cycles)                                      # first use: the runtime binds the stream to a hardware queue here
        alone = _spin_ms(torch, chosen, cycles)
        while alone < 2.0 and cycles < (1 << 34):                            # a spin of a few ms: well above launch overhead and host timer noise
            cycles *= 4
            alone = _spin_ms(torch, chosen, cycles)
        verified = True
        while len(chosen) < n:
            pick = cand = None
            for _ in range(tries):
                cand = torch.cuda.Stream(device=device)                       # the next stream of torch's pool
                if any(cand.cuda_stream == s.cuda_stream for s in chosen):
                    continue
                _spin_ms(torch, [cand], cycles // 16)                         # bind
                together = min(_spin_ms(torch, chosen + [cand], cycles) for _ in range(2))
                if together < 1.5 * alone:                                    # side by side: ~ 1 x; one queue: (len + 1) x
                    pick = cand
                    break
            verified = verified and pick is not None
            if pick is None:                                                  # nothing verified: at least never the SAME stream twice (round-5 advice) -- and say so
                import warnings
                pick = cand
                while pick is None or any(pick.cuda_stream == s.cuda_stream for s in chosen):
                    pick = torch.cuda.Stream(device=device)
                warnings.warn("group_streams: no stream was seen to overlap the chosen ones (one hardware queue?): scene groups may run one after the other", RuntimeWarning)
            chosen.append(pick)
        return chosen, verified
