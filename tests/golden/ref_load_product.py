"""BUILD CONTAINER ONLY (imports `/root/reference`): the REFERENCE restores a checkpoint and replay files written by
the product on an MI355X (`tests/golden/product_ckpt/`, produced by
`tests/test_surface_parity_gpu.py::test_product_writes_files_for_the_reference`), acts and keeps training.
Run by `tests/test_interop_cpu.py` in a subprocess (the reference's package is also called `algorithm`)."""
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import ref_shims  # noqa: E402

ref_shims.install()
from algorithm.sac_base import SAC_Base  # noqa: E402
from algorithm.utils.enums import SEQ_ENCODER  # noqa: E402
from make_golden import load_ref_nn, seed_all  # noqa: E402  (also switches the replay to its synchronous drive)


def main():
    src = HERE / 'product_ckpt'
    tmp = Path(tempfile.mkdtemp())
    (tmp / 'model').mkdir()
    for name in ('3.pth', '3-rb_tree.npy', '3-rb_storage.npz'):
        shutil.copy(src / name, tmp / 'model' / name)
    expect = np.load(src / 'expect.npz')
    seed_all(0)
    torch.set_num_threads(1)
    sac = SAC_Base(obs_names=['vector'], obs_shapes=[(6,)], d_action_sizes=[], c_action_size=2, model_abs_dir=tmp,
                   nn=load_ref_nn('envs/test/nn_rnn.py'), device='cpu', batch_size=16, n_step=3, burn_in_step=2,
                   seq_encoder=SEQ_ENCODER.RNN, replay_config={'capacity': 128})
    assert sac.get_global_step() == 3, sac.get_global_step()
    rb = sac.replay_buffer
    assert np.array_equal(rb._sum_tree._tree.view(np.uint32), expect['tree'].view(np.uint32)), 'tree bytes'
    assert rb.size == 128 and rb._trans_storage._id == 165
    assert set(rb._trans_storage._buffer) >= {'_id', 'index', 'obs_vector', 'action', 'reward', 'done', 'mu_prob',
                                              'pre_seq_hidden_state', 'last_mask'}
    np.testing.assert_allclose(sac.log_c_alpha.detach().numpy(), expect['log_c_alpha'], rtol=0, atol=0)
    for opt in (sac.optimizer_rep, sac.optimizer_policy, *sac.optimizer_q_list):
        steps = {float(s['step']) for s in opt.state.values()}
        assert steps == {3.0}, steps           # Adam moments and step counts came across
    a, p, h = sac.choose_action([expect['obs']], np.zeros((2, 2), np.float32),
                                np.zeros((2, *sac.seq_hidden_state_shape), np.float32), disable_sample=True)
    np.testing.assert_allclose(a, expect['action'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(p, expect['prob'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(h, expect['hidden'], rtol=1e-5, atol=2e-6)
    assert sac.train() == 4                       # ... and the reference keeps training on the product's replay
    assert np.isfinite(rb._sum_tree._tree).all()
    sac.close()
    shutil.rmtree(tmp)
    print('reference restored the product files: ok')


if __name__ == '__main__':
    main()
