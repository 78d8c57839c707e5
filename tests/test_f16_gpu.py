"""The IEEE-half build (BASELINE.json configs[4]: "run/ctc/cnn ... fp16 MFMA"): libasr_hip_f16.so is the same kernels compiled with
-DASR_ACT_F16 (csrc/common.hpp: conversions and the MFMA instruction are the only things that differ), selected per process by
ASR_ACT=f16.  The activation format is a property of the loaded library, so the half build is tested in a CHILD process: one pytest
run over the kernels and recipes of the convolutional path (the GRU / SRU recurrences refuse in this build: DESIGN.md 13.9), with the
rounding-matched oracle rounding to float16 as well (oracle/bf16.py ACT), plus the loss-scaling tests below, which only run there."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from asr._lib import act_dtype as _act_dtype

from oracle import model as omodel

pytestmark = pytest.mark.gpu
HALF = _act_dtype() is torch.float16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the child process runs: every kernel test of the convolutional path and every recipe test of run/ctc/cnn -- ONE pytest run
FILES = ["tests/test_kernels_gpu.py", "tests/test_nn_gpu.py", "tests/test_model_gpu.py", "tests/test_ctc_gpu.py", "tests/test_f16_gpu.py"]
SELECTION = ("(gemm or conv or maxout or layernorm or colsum or cast_transpose or pack_input or layer_stack or clip_decay or non_finite or step_control "
             "or activations or crelu or glu or weightnorm or upsampling or batch_normalization or generic_layout or gaussian or handover "
             "or projection_bias or cnn_recipes or first_block or half_build_only) and not gru and not sru and not recurrence")


def test_the_half_build_over_the_convolutional_path_in_a_child_process(device):
    if HALF:
        pytest.skip("this IS the child process")
    env = dict(os.environ, ASR_ACT="f16")
    env.pop("ASR_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-rs"] + FILES + ["-k", SELECTION],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=2400)
    tail = r.stdout.decode(errors="replace")[-8000:]
    assert r.returncode == 0, "half build:\n" + tail
    last = tail.strip().splitlines()[-1]
    assert " passed" in last and "failed" not in last, tail
    print("half build:", last)


def _recipe(device, arch="zhang+residual", nconv=4, seed=0):
    from asr.model import cnn
    from asr.model.architectures import build_model
    torch.manual_seed(seed)
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = 19, 3, 16, 24, nconv, arch
    model = build_model(cfg).to_gpu()
    x, labels, x_len, l_len = omodel.synthetic_batch(3, 40, 19, Lmin=2, Lmax=6, seed=5)
    batch = tuple(t.to(device) for t in (x, labels, x_len, l_len))
    with torch.no_grad():
        model(batch[0])
    return cfg, model, batch


@pytest.mark.skipif(not HALF, reason="runs in the child process of the test above (ASR_ACT=f16)")
def test_half_build_only_library_and_refusals(device):
    from asr import _lib, _ops
    assert os.path.basename(_lib.lib()._name) == "libasr_hip_f16.so" and _lib.lib().asr_act_dtype() == 1
    assert _ops.BF16 is torch.float16
    H, T, B = 64, 4, 2
    with pytest.raises(_lib.AsrHipError):       # the recurrences are bfloat16-only (their saved-gate image is already half: DESIGN 13.9)
        _ops.gru_fwd(torch.zeros(T * B, 3 * H, device=device), torch.zeros(1, 3 * H, H, device=device, dtype=torch.float16),
                     torch.zeros(3 * H, device=device), T, B, H, 1)


@pytest.mark.skipif(not HALF, reason="runs in the child process of the test above (ASR_ACT=f16)")
def test_half_build_only_loss_scaling_keeps_the_small_gradients(device):
    """Activation gradients below the half format's normal range (6.1e-5) lose bits, below 6e-8 they vanish.  The full-size step's are
    ~1e-6 .. 1e-4 per element (mean CTC loss over 32 utterances x 1000 frames); this toy recipe's are ~0.05, so the test seeds its backward
    pass with 2^-16 to put them where the full-size ones are, and with 2^-16 x 1024 for the scaled run: divided by the seed again, the
    parameter gradients of the scaled run agree with the float32 oracle as well as half's 11 bits allow, the unscaled run's do not --
    the reason Optimizer.loss_scaling exists."""
    from asr.loss import connectionist_temporal_classification
    from asr.functions import join_side_stream
    from oracle import cnn as ocnn
    cfg, model, (xd, ld, xl, ll) = _recipe(device)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    out = ocnn.forward(cfg.architecture, cfg, params, xd.cpu(), matched=False, fused_logit_bias=False)
    omodel.ctc_mean_loss(ocnn.logits_tbv(out), ld.cpu(), xl.cpu(), ll.cpu()).backward()

    def worst(seed):
        for p in model.parameters():
            p.grad = None
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        loss.backward(gradient=torch.full_like(loss, seed))
        join_side_stream()
        torch.cuda.synchronize()
        errs = {}
        for n, p in model.named_parameters():
            a, b = (p.grad.double().cpu() / seed).flatten(), params[n].grad.double().flatten()
            errs[n] = float((a - b).norm() / (b.norm() + 1e-30))
        return max(errs.values()), max(errs, key=errs.get)

    scaled, unscaled = worst(2.0 ** -6), worst(2.0 ** -16)
    print("half build, worst parameter gradient against the float32 oracle: small gradients x 1024 %.2e (%s), as they are %.2e (%s)" % (scaled + unscaled))
    assert scaled[0] < 0.15, scaled                 # (0.25: the bfloat16 build's bar against the float32 oracle in test_model_gpu)
    assert unscaled[0] > 2 * scaled[0], (scaled, unscaled)


@pytest.mark.skipif(not HALF, reason="runs in the child process of the test above (ASR_ACT=f16)")
def test_half_build_only_dynamic_loss_scale_backs_off_and_grows(device):
    """asr_step_control_scaled on the device: an overflowing seed drops the step and halves S until the backward pass fits (parameters
    untouched meanwhile), `interval` applied steps in a row double it; the update itself is the unscaled one (compared with a static
    scale on a copy of the model)."""
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping
    cfg, model, (xd, ld, xl, ll) = _recipe(device, seed=1)
    opt = Adam(alpha=1e-3)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    opt.loss_scaling(interval=3)
    opt._loss_scaling = (3, 2.0 ** 24)          # start where the very first backward pass overflows the half range
    before = None
    drops = 0
    for step in range(40):
        opt.update(lossfun=lambda: connectionist_temporal_classification(model(xd), ld, 0, xl, ll))
        if before is None:
            before = opt.flat_parameters().clone()
        s, overflows = opt.loss_scale()
        if opt.applied_steps() == 0:
            assert torch.equal(opt.flat_parameters(), before)       # dropped steps leave parameters and moments alone
            drops = overflows
        elif opt.applied_steps() >= 7:
            break
    s, overflows = opt.loss_scale()
    assert drops >= 1 and overflows >= drops and opt.applied_steps() >= 7
    assert opt.t == opt.applied_steps() + overflows                 # every attempted step is either applied or counted as an overflow
    assert 0.0 < s < 2.0 ** 24 and np.log2(s) == int(np.log2(s))
    print("dynamic loss scale: %d overflow(s) from 2^24, settled at 2^%d after %d applied steps" % (overflows, int(np.log2(s)), opt.applied_steps()))

    # the applied update does not depend on S (up to the roundings S moves): a static 256 and a static 4096 give the same parameters
    results = []
    for scale in (256.0, 4096.0):
        cfg2, model2, _ = _recipe(device, seed=2)
        o = Adam(alpha=1e-3)
        o.setup(model2)
        o.add_hook(GradientClipping(1.0))
        o.loss_scaling(scale=scale)
        p0 = None
        for _ in range(3):
            o.update(lossfun=lambda: connectionist_temporal_classification(model2(xd), ld, 0, xl, ll))
            if p0 is None:
                p0 = o.flat_parameters().clone()      # after one step (the flat buffer exists from the first update on)
        assert o.applied_steps() == 3 and o.loss_scale() == (scale, 0)
        results.append((o.flat_parameters().clone(), p0))
    moved = (results[0][0] - results[0][1]).abs().max().item()
    apart = (results[0][0] - results[1][0]).abs().max().item()
    assert moved > 1e-4 and apart < 0.1 * moved, (moved, apart)
