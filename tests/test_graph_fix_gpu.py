"""GPU: the fix-up pass over captured graphs (`asac_graph_replace_memset_nodes`, csrc/graph_fix.hip).  On ROCm 7.2 / gfx950 a
captured hipMemsetAsync of >= 16 bytes takes effect on the first launch of the instantiated graph only, and ATen zeroes the
semaphores of its split reductions with one: a captured `x.sum(0)` over thousands of rows (every nn.Linear's bias gradient in
a captured train step) is wrong from the second replay on.  The pass turns memset nodes into kernel nodes; here: byte-exact
fills at every alignment / width / element size, over several replays, and the reduction itself."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hip():
    hip = ctypes.CDLL('libamdhip64.so')
    for name in ('hipMemsetAsync', 'hipMemsetD16Async', 'hipMemsetD32Async'):
        getattr(hip, name).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    return hip


def _capture(body):
    from asac_amd import native
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph, stream=side):
        body()
    counts = native.graph_replace_memset_nodes(int(graph.raw_cuda_graph()))
    graph.instantiate()
    return graph, counts


def test_memset_nodes_become_fills_byte_exact_over_replays():
    import asac_amd  # noqa: F401
    hip = _hip()
    buf = torch.zeros(1 << 18, dtype=torch.uint8, device='cuda')
    cases = []       # (offset, bytes, element size, value)
    rng = np.random.default_rng(0)
    off = 64
    for es, fn in ((1, 'hipMemsetAsync'), (2, 'hipMemsetD16Async'), (4, 'hipMemsetD32Async')):
        for count in (1, 3, 4, 5, 16, 17, 63, 64, 257, 1000, 4099):
            for mis in {1: (0, 1, 3, 7), 2: (0, 2, 6), 4: (0, 4, 12)}[es]:
                start = off + mis
                cases.append((start, count * es, es, int(rng.integers(1, 2 ** (8 * es))), fn, count))
                off = (start + count * es + 48 + 15) // 16 * 16
    assert off < buf.numel()

    def body():
        buf.add_(0)       # a kernel in front of the memsets
        for start, _, _, value, fn, count in cases:
            assert getattr(hip, fn)(buf.data_ptr() + start, value, count, torch.cuda.current_stream().cuda_stream) == 0
        buf.add_(0)

    graph, (replaced, kept) = _capture(body)
    assert replaced == len(cases) and kept == 0
    want = np.full(buf.numel(), 0xA5, dtype=np.uint8)
    for start, nbytes, es, value, _, count in cases:
        want[start:start + nbytes] = np.frombuffer(np.array([value], dtype={1: np.uint8, 2: np.uint16, 4: np.uint32}[es]).tobytes() * count, dtype=np.uint8)
    for _ in range(4):
        buf.fill_(0xA5)       # everything a fill does not own must survive; everything it owns is rewritten every replay
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), want)


@pytest.mark.parametrize('rows,cols', [(9216, 64), (9216, 192), (20736, 64), (73728, 8)])
def test_captured_split_reduction_replays_correctly_after_the_pass(rows, cols):
    import asac_amd  # noqa: F401
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device='cuda')
    y = torch.zeros(cols, device='cuda')

    def body():
        y.copy_((x * 1.0).sum(0))

    graph, _ = _capture(body)
    for it in range(6):
        x.copy_(torch.randn(rows, cols, device='cuda'))
        want = x.double().sum(0).float()
        if it % 2:
            torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose(y.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_train_step_graph_reports_its_memset_nodes():
    """the learner's capture runs the pass (cfg_attn_h64's representation holds nn.Linear projections over 9 216 rows)"""
    import bench
    from tests import parity_utils as pu
    from algorithm.sac_base import SAC_Base
    from algorithm.utils.enums import SEQ_ENCODER
    cfg = bench.CONFIGS['cfg_attn_h64']
    torch.manual_seed(0)
    agent = SAC_Base(cfg['obs_names'], cfg['obs_shapes'], [], cfg['c_action_size'], None, pu.plugin(cfg['plugin']), device='cuda:0',
                     seq_encoder=SEQ_ENCODER.ATTN, n_step=cfg['n_step'], burn_in_step=cfg['burn_in_step'], batch_size=64,
                     replay_config={'capacity': 4096}, hip_config={'use_graph': True, 'graph_warmup': 1})
    rng = np.random.default_rng(1)
    for _ in range(4):
        agent.put_episode(**pu.synthetic_episode(rng, cfg['obs_shapes'], [], cfg['c_action_size'], cfg['hidden'], 60))
    for _ in range(3):
        agent.train()
    assert agent._graph is not None
    replaced, kept = agent._graph_memsets
    assert kept == 0
    agent.close()


def test_two_dimensional_memset_nodes_are_left_alone_and_reported():
    """the pass rewrites 1-D memsets only (what ATen and the runtime issue inside a train step); a pitched 2-D memset stays a
    memset node and is counted, so that a learner can log it"""
    import asac_amd  # noqa: F401
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemset2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                     ctypes.c_void_p]
    buf = torch.full((64, 256), 7, dtype=torch.uint8, device='cuda')

    def body():
        buf.add_(0)
        assert hip.hipMemset2DAsync(buf.data_ptr(), 256, 3, 100, 16, torch.cuda.current_stream().cuda_stream) == 0
        buf.add_(0)

    graph, (replaced, kept) = _capture(body)
    assert (replaced, kept) == (0, 1)
    graph.replay()
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert (got[:16, :100] == 3).all() and (got[:16, 100:] == 7).all() and (got[16:] == 7).all()
