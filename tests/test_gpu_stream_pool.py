"""stream_pool.concurrent_streams: the streams it hands out run beside each other and beside the caller's stream -- two spin
kernels on two of them take as long as one -- which arbitrary pooled streams of torch do not (four hardware queues)."""

import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair_ms(a, b, cycles=2_000_000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def test_streams_handed_out_run_beside_each_other_and_the_caller():
    from hopwise_amd.stream_pool import concurrent_streams

    dev = torch.device("cuda", 0)
    junk = [torch.cuda.Stream() for _ in range(5)]  # whatever the process created before
    for s in junk:
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    main = torch.cuda.current_stream(dev)
    side = concurrent_streams(3, dev)
    assert len(side) == 3 and len({s.cuda_stream for s in side} | {main.cuda_stream}) == 4
    assert concurrent_streams(3, dev) is side  # cached: same streams for the same request
    _pair_ms(side[0], side[0])
    single = min(_pair_ms(side[0], side[0]) for _ in range(2)) / 2
    for a, b in ((side[0], side[1]), (side[0], side[2]), (side[1], side[2]), (main, side[0]), (main, side[1]), (main, side[2])):
        _pair_ms(a, b)
        assert min(_pair_ms(a, b) for _ in range(2)) < 1.5 * single, "two handed-out streams share a hardware queue"
    # a stream that runs beside a given one (the lone PGPR search's action stream)
    other = concurrent_streams(1, dev, beside=[side[1]])[0]
    assert other.cuda_stream != side[1].cuda_stream
    _pair_ms(other, side[1])
    assert min(_pair_ms(other, side[1]) for _ in range(2)) < 1.5 * single
    # more streams than queues: still that many distinct streams
    many = concurrent_streams(7, dev)
    assert len({s.cuda_stream for s in many}) == 7
