"""Host-side waiting of the batch mode on a real device (unires_amd/_host.py): stream marks
(`unires_mark_*`, include/unires_hip.h), the pacer of the ADMM loop and the sleeping wait in front of a
read-back.  Nothing here has a counterpart in the reference (single process, unires/run.py) - what is tested is
that the marks tell the truth about the stream."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _busy(dev, n=24):
    """~ tens of milliseconds of device work on the current stream."""
    a = torch.rand((4096, 4096), device=dev)
    for _ in range(n):
        a = (a @ a).clamp_(0, 1)
    return a


def test_stream_mark_follows_the_stream(dev):
    from unires_amd._host import StreamMark
    with torch.cuda.device(dev):
        torch.cuda.synchronize()
        m = StreamMark()
        assert m.reached(0) and not m.reached(1)
        a = _busy(dev)
        ev = torch.cuda.Event()
        ev.record()
        v = m.signal()
        assert v == 1
        seen_before_event = m.reached(v) and not ev.query()  # the mark must never run ahead of the stream
        assert not seen_before_event
        m.wait(v)
        assert m.reached(v) and ev.query()
        # values count up along the stream; a later value implies the earlier ones
        v2, v3 = m.signal(), m.signal()
        m.wait(v3)
        assert (v2, v3) == (2, 3) and m.reached(v2)
        del a


def test_marks_on_two_streams_are_independent(dev):
    from unires_amd._host import StreamMark
    with torch.cuda.device(dev):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        m1, m2 = StreamMark(), StreamMark()
        with torch.cuda.stream(s1):
            a = _busy(dev, 48)
            ev = torch.cuda.Event()
            ev.record(s1)
            v1 = m1.signal(s1)
        with torch.cuda.stream(s2):
            v2 = m2.signal(s2)  # nothing in front of it on this stream
        m2.wait(v2)
        assert m2.reached(v2)
        m1.wait(v1)
        assert ev.query()  # (the stream itself may still be retiring the mark's own kernel)
        del a


def test_pacer_bounds_the_hosts_lead_and_wait_blocking_leaves_nothing_to_wait_for(dev):
    from unires_amd._host import Pacer, wait_blocking
    with torch.cuda.device(dev):
        torch.cuda.synchronize()
        p = Pacer(1)
        done = []
        for k in range(5):
            a = _busy(dev, 8)
            ev = torch.cuda.Event()
            ev.record()
            done.append(ev)
            p.step()
            assert len(p._pending) <= 1
            if k >= 1:
                assert done[k - 1].query()  # step k returned: step k - 1 has finished on the device
        p.drain()
        assert done[-1].query() and not p._pending
        a = _busy(dev, 16)
        t0 = time.perf_counter()
        ev = torch.cuda.Event()
        ev.record()
        wait_blocking(dev)
        assert ev.query()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        assert time.perf_counter() - t1 < 0.5 * max(t1 - t0, 1e-3) + 1e-3  # the synchronisation found nothing left
        del a


def test_admm_loop_runs_with_and_without_the_pacer(dev):
    """Same iterates whatever the pacing (it only ever waits)."""
    import unires_amd as U
    from tests.helpers import gpu_structs, make_problem
    outs = []
    for pace in (0, 1, 2):
        prob = make_problem(seed=3, dim_y=(20, 18, 22), n_channels=2, thick=3, regime='sr', rot=0.1, trans=1.0)
        xg, yg, sett = gpu_structs(prob, dev)
        sett.host_pace = pace
        sett.cgs_max_iter = 4
        z, w = prob['z'].clone().to(dev), prob['w'].clone().to(dev)
        tmp = torch.zeros_like(yg[0].dat)
        for it in range(4):
            yg, z, w, tmp, _ = U._update_admm(xg, yg, z, w, float(prob['rho']), tmp, None, it, sett)
        outs.append([yc.dat.clone() for yc in yg])
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)


def test_a_mark_that_cannot_come_raises_instead_of_hanging(dev):
    """The wait's watchdog: a value nobody signalled, on a stream that has drained -> RuntimeError after ~1 s."""
    from unires_amd._host import StreamMark
    with torch.cuda.device(dev):
        m = StreamMark()
        v = m.signal()
        m.wait(v)
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match='without reaching its mark'):
            m.wait(v + 3)
        assert 0.9 < time.perf_counter() - t0 < 5.0
