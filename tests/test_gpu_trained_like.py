"""Parity on a TRAINED-LIKE net (tests/trained_fixture.py): the He-init + calibrated-heads fixtures of the other GPU tests
have 3 % of their pixels inside the bf16 noise band and gradient tensors that move by 1e-2 with the summation order -- here the same checks run
on a net whose fused loss has fallen 18x on learnable synthetic frames (fused logits: std 8-12, 0.5 % of the pixels within |logit| < 0.5), i.e.
with margins like a real checkpoint's.  Reference being matched: networks/vgg_osvos.py:59-74 + layers/osvos_layers.py:19-48 and their autograd,
restated in oracle/torch_ref.py and run in float64 on the CPU (train_online.py:181-187 thresholds the fused logits at 0 -> IoU)."""
import json
import os

import numpy as np
import pytest
import torch

import trained_fixture as tf

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_like.json")


_ORACLE_CACHE = {}      # float64 oracle results shared by the parametrisations of a test (host time, not GPU time)


@pytest.fixture(scope="module")
def trained():
    wts, frames, curve = tf.train_like()
    return wts, frames, curve


def _iou(a, b):
    a, b = a > 0, b > 0
    u = np.logical_or(a, b).sum()
    return 1.0 if u == 0 else float(np.logical_and(a, b).sum()) / float(u)


_ORACLE = {}


def _oracle(wts, x, m, key):
    """float64 torch-CPU oracle: logits, per-head losses, every gradient of 0.5 * side losses + fused loss"""
    from oracle import torch_ref
    if key not in _ORACLE:
        p = torch_ref.as_leaf_params(wts, dtype=torch.float64)
        xi = torch.from_numpy(x).double().requires_grad_()
        outs = torch_ref.forward(p, xi)
        losses = [torch_ref.cbce_loss(o, torch.from_numpy(m).double(), size_average=False) for o in outs]
        (0.5 * sum(losses[:-1]) + losses[-1]).backward()
        grads = {k: v.grad.clone() for k, v in p.items() if v.grad is not None and not k.startswith("upscale")}
        grads["input"] = xi.grad.clone()
        _ORACLE[key] = ([o.detach().numpy() for o in outs], [float(l) for l in losses], grads)
    return _ORACLE[key]


def _gpu(wts, x, m, precision):
    from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
    net = tf.build(wts, precision)
    xg = torch.from_numpy(x).requires_grad_()
    outs = net.forward(xg.cuda())
    gt = torch.from_numpy(m).cuda()
    losses = [cbce(o, gt, size_average=False) for o in outs]
    (0.5 * sum(losses[:-1]) + losses[-1]).backward()
    grads = {k: v.grad.cpu().double() for k, v in net.named_parameters() if v.grad is not None}
    grads["input"] = xg.grad.double()
    return [o.detach().cpu().double().numpy() for o in outs], [float(l.item()) for l in losses], grads


def _cases(frames):
    from oracle import synth
    h, w = tf.RECIPE["h"], tf.RECIPE["w"]
    return [("train0", frames[0][0], frames[0][1]), ("train3", frames[3][0], frames[3][1]),
            ("heldout",) + synth.trainable_frame(1, h, w, seed=tf.RECIPE["frame_seed"] + 99),
            ("heldout_240x427",) + synth.trainable_frame(1, 240, 427, seed=tf.RECIPE["frame_seed"] + 98)]


def test_fixture_reaches_its_pinned_loss_curve(trained):
    """the regenerated fixture IS the pinned one: same loss curve (fp32 summation order may differ between runs and boxes), fused loss down by
    more than 10x, margins as recorded by tools/make_trained_fixture.py (which also pins the first optimizer steps against the float64 oracle)"""
    wts, frames, curve = trained
    first, last = curve[0][1], curve[-1][1]
    assert first / last > 10.0, (first, last)
    assert all(np.isfinite(v).all() for v in wts.values())
    if not os.path.exists(GOLDEN):
        pytest.skip("tests/golden/trained_like.json not generated yet (tools/make_trained_fixture.py --write on a GPU box)")
    g = json.load(open(GOLDEN))
    assert g["recipe"] == tf.RECIPE
    pinned = dict((int(s), l) for s, l in g["curve"])
    for s, l in curve:
        assert abs(l - pinned[s]) <= 0.03 * pinned[s], (s, l, pinned[s])
    # the float64 oracle's replay of the first 30 optimizer steps (same frames, same schedule), recorded by the generator: the first steps --
    # the calibrated He-init net is steep, the loss jumps 3.1e3 -> 8.2e3 -> 1.6e3 within ten steps -- agree to 1e-4, later ones to 1-2 %
    # (a perturbation of one fp32 ulp grows that far in this phase; the QUIET phase is what test_twenty_optimizer_steps... checks at 2e-4)
    ref = dict((int(s), l) for s, l in g["oracle"]["ref_curve"])
    assert abs(curve[0][1] - ref[1]) <= 1e-5 * ref[1]
    for s, l in curve:
        if s in ref:
            assert abs(l - ref[s]) <= 2e-2 * ref[s], (s, l, ref[s])
    assert g["oracle"]["max_rel_loss_diff"] < 5e-2


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x3b2", "fp32x3h2", "fp32h2"])
def test_fp32_arithmetics_meet_the_flat_survey_bars_on_the_trained_like_net(trained, precision):
    """SURVEY 8(d) bars with NO escape hatch (no check_grad_either, no 'or 2x the reference's own distance') on two training frames, a
    held-out frame and a held-out frame at four times the pixels, against the float64 oracle:
      logits <= 1e-3 std on all five heads (measured <= 6e-6), losses <= 1e-5 relative (<= 3e-6), mask IoU(logit > 0) >= 1 - 1e-3 (1.000000),
      all parameter gradients as ONE vector <= 1e-3 relative L2 (<= 3.7e-4), every parameter gradient tensor <= 2e-3 (<= 1.2e-3; 58 of 60
      tensor x case pairs below 8e-4 -- what is left are single ReLU / arg-max flips of float32 arithmetic, which ANY float32 implementation
      has against float64), the input gradient -- computed by the reference, used by nothing -- <= 1e-2 (1e-5 .. 7e-3, the flip-iest tensor)."""
    # 'fp32h2' (FP16 pairs in the FORWARD too; round 6): logits, losses and IoU at the flat bars (its activations are closer to float64 than fp32x3's and it
    # flips no more ReLUs: tools/net_flip_probe.py, profiles/r06_fp32h2.txt) -- but WHICH pre-activations within 1e-6 std of zero flip is a lottery, one
    # flip in a 30 x 54-pixel layer moves the gradient vector by ~1e-3, and on this fixture this forward draws 1.4e-3 (held-out frame) where fp32x3
    # draws 5e-4 and the exact fp32 kernels 1e-4: its gradient bars are 3x the flat ones, and it is not a default of anything.
    gbar = 3.0 if precision == "fp32h2" else 1.0
    wts, frames, _ = trained
    worst = {"logit": 0.0, "loss": 0.0, "iou": 1.0, "grad": (0.0, "")}
    for name, x, m in _cases(frames):
        t_outs, t_losses, t_grads = _oracle(wts, x, m, name)
        outs, losses, grads = _gpu(wts, x, m, precision)
        for i in range(5):
            e = float(np.abs(outs[i] - t_outs[i]).max() / t_outs[i].std())
            worst["logit"] = max(worst["logit"], e)
            assert e <= 1e-3, (name, i, e)
            le = abs(losses[i] - t_losses[i]) / abs(t_losses[i])
            worst["loss"] = max(worst["loss"], le)
            assert le <= 1e-5, (name, i, le)
        j = _iou(outs[4], t_outs[4])
        worst["iou"] = min(worst["iou"], j)
        assert j >= 1 - 1e-3, (name, j)
        errs = sorted(((float((grads[k] - truth).norm() / (truth.norm() + 1e-300)), k, truth.numel()) for k, truth in t_grads.items()), reverse=True)
        print("   %s %s worst gradients:" % (precision, name), [("%s[%d]" % (k, n), "%.1e" % e) for e, k, n in errs[:6]])
        num = sum(float((grads[k] - t).norm() ** 2) for k, t in t_grads.items() if k != "input")
        den = sum(float(t.norm() ** 2) for k, t in t_grads.items() if k != "input")
        print("   %s %s all parameter gradients as one vector: %.2e" % (precision, name, (num / den) ** 0.5))
        assert (num / den) ** 0.5 <= gbar * 1e-3, (name, (num / den) ** 0.5)
        for e, k, n in errs:
            if e > worst["grad"][0]:
                worst["grad"] = (e, name + ":" + k)
            assert e <= gbar * (1e-2 if k == "input" else 2e-3), (name, k, e)
    print("trained-like %s vs float64: max |dlogit| %.2e std, loss rel %.2e, min IoU %.6f, worst gradient %.2e (%s)"
          % (precision, worst["logit"], worst["loss"], worst["iou"], worst["grad"][0], worst["grad"][1]))


def _bf16_rows(trained):
    wts, frames, _ = trained
    rows = []
    for name, x, m in _cases(frames):
        t_outs, t_losses, t_grads = _oracle(wts, x, m, name)
        outs, losses, grads = _gpu(wts, x, m, "bf16")
        e_logit = [float(np.abs(outs[i] - t_outs[i]).max() / t_outs[i].std()) for i in range(5)]
        e_loss = [abs(losses[i] - t_losses[i]) / abs(t_losses[i]) for i in range(5)]
        j = _iou(outs[4], t_outs[4])
        fl = (outs[4] > 0) != (t_outs[4] > 0)
        rms = float(np.sqrt(np.mean((outs[4] - t_outs[4]) ** 2)))
        flips = (int(fl.sum()), int((fl & (np.abs(t_outs[4]) > 4.0 * rms)).sum()))      # (all, outside the |logit| <= 4 rms(dlogit) band)
        ge = sorted(((float((grads[k] - t).norm() / (t.norm() + 1e-300)), k) for k, t in t_grads.items()), reverse=True)
        print("   bf16 %s worst gradients:" % name, [(k, "%.2f" % e) for e, k in ge[:8]])
        num = sum(float((grads[k] - t).norm() ** 2) for k, t in t_grads.items() if k != "input")
        den = sum(float(t.norm() ** 2) for k, t in t_grads.items() if k != "input")
        print("   bf16 %s all parameter gradients as one vector: %.3f" % (name, (num / den) ** 0.5))
        rows.append((name, e_logit, e_loss, j, flips, ge, (num / den) ** 0.5))
        print("trained-like bf16 %s: max |dlogit| / std per head %s | loss rel %s | fused IoU %.6f (%d flipped of %d, %d outside the noise band) | gradients worst %.3f (%s) median %.3f"
              % (name, ["%.3f" % e for e in e_logit], ["%.1e" % e for e in e_loss], j, flips[0], outs[4].size, flips[1], ge[0][0], ge[0][1], ge[len(ge) // 2][0]))
    return rows


def test_bf16_on_the_trained_like_net_measured_bars(trained):
    """Where bf16 (bf16 MFMA operands + bf16 trunk tensors) lands when the margins are real, asserted at what was MEASURED plus a small margin, so
    that a regression cannot hide inside a relaxed bar (profiles/r04_trained_like.txt, re-measured in round 5):
      max |dlogit| <= 0.1 std (flat SURVEY bar): met on every head (0.007-0.020); asserted at 0.03;
      mask IoU: 0.99869 .. 0.99981 over the four cases at the round-5 tree (1-15 flipped pixels, every one of them among the ~0.1 % of pixels with
        |logit| < 0.1 of a std-7..12 map; round 4's tile rules gave 0.99888 .. 0.99984 -- a different fp32 summation order moves single pixels, and one
        pixel is 1.9e-4 of IoU at 120x214); asserted >= 0.9985 AND every flipped pixel inside the |logit| <= 4 rms(dlogit) band;
      gradients: all parameter gradients as one vector 0.05-0.10, asserted <= 0.15; trunk / side_prep / fuse tensors <= 0.25 measured, asserted 0.3;
      loss: 5e-5 .. 1.3e-2 relative (the loss has fallen 23x and what is left sits on the few uncertain pixels); asserted <= 2.5e-2.
    The FLAT bars of SURVEY 8(d) / north_star that bf16 does not meet on this fixture are the next test (xfail), not a looser number here."""
    for name, e_logit, e_loss, j, flips, ge, one_vec in _bf16_rows(trained):
        assert max(e_logit) <= 0.03, (name, e_logit)
        assert j >= 0.9985 and flips[1] == 0, (name, j, flips)
        assert one_vec <= 0.15, (name, one_vec)
        for e, k in ge:
            if k.startswith(("stages.", "side_prep.", "fuse.")):
                assert e <= 0.3, (name, k, e)
        assert max(e_loss) <= 2.5e-2, (name, e_loss)


def test_fp32x2_on_the_trained_like_net(trained):
    """Precision 'fp32x2' (round 6: the f32x3 kernels with TWO bf16 pieces per operand, three MFMA products per fp32 product; OSVOS_FLAG_X3_TWO_PIECES) on
    the net with real margins, against the float64 oracle (vgg_osvos.py:59-74, osvos_layers.py:19-48 and their autograd): logits <= 1e-3 std -- the
    flat fp32 bar -- on every head, mask IoU >= 1 - 1e-3, losses <= 1e-4 relative (the flat fp32 bar is 1e-5: NOT promised, the emulation of the mode
    puts single heads at 1.7e-5), all parameter gradients as one vector <= 5e-3; and the mode really is another arithmetic than 'fp32x3'."""
    wts, frames, _ = trained
    rows = []
    for name, x, m in _cases(frames):
        t_outs, t_losses, t_grads = _oracle(wts, x, m, name)
        outs, losses, grads = _gpu(wts, x, m, "fp32x2")
        o3, _, _ = _gpu(wts, x, m, "fp32x3")
        e_logit = max(float(np.abs(outs[i] - t_outs[i]).max() / t_outs[i].std()) for i in range(5))
        e_loss = max(abs(losses[i] - t_losses[i]) / abs(t_losses[i]) for i in range(5))
        j = _iou(outs[4], t_outs[4])
        num = sum(float((grads[k] - t).norm() ** 2) for k, t in t_grads.items() if k != "input")
        den = sum(float(t.norm() ** 2) for k, t in t_grads.items() if k != "input")
        one_vec = (num / den) ** 0.5
        rows.append((name, e_logit, e_loss, j, one_vec))
        assert e_logit <= 1e-3 and j >= 1 - 1e-3 and e_loss <= 1e-4 and one_vec <= 5e-3, rows[-1]
        assert any(not np.array_equal(outs[i], o3[i]) for i in range(5)), name
    print("trained-like fp32x2 vs float64 (case, max |dlogit| / std, worst loss rel, fused IoU, gradients as one vector):",
          [(n, "%.1e" % a, "%.1e" % b, "%.6f" % c, "%.1e" % d) for n, a, b, c, d in rows])


@pytest.mark.xfail(strict=False, reason="bf16 on the trained-like fixture sits AT north_star's IoU line (0.99869 .. 0.99981 vs 1 - 1e-3: 7 flipped "
                                        "pixels of 25680 on two cases) and above SURVEY 8(d)'s 2e-3 loss bar (up to 1.3e-2): recorded as a known miss, "
                                        "the measured bars are asserted by the test above.  profiles/r06_bf16_error_budget.txt: no per-stage "
                                        "mixed-precision policy under +30 % step time brings the loss under 2e-3 on this fixture (only all-layer operand "
                                        "splits do), so this stays a recorded miss of bf16 -- 'fp32x3' meets every flat bar")
def test_bf16_on_the_trained_like_net_flat_survey_bars(trained):
    """The flat bars: mask IoU >= 1 - 1e-3 (north_star), loss <= 2e-3, logits <= 0.1 std, gradients (one vector) <= 0.25."""
    for name, e_logit, e_loss, j, flips, ge, one_vec in _bf16_rows(trained):
        assert max(e_logit) <= 0.1 and one_vec <= 0.25, (name, e_logit, one_vec)
        assert j >= 1 - 1e-3, (name, j, flips)
        assert max(e_loss) <= 2e-3, (name, e_loss)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x3b2", "fp32h2", "fp32x3h2"])
def test_window_fused_pass_equals_the_sequential_micro_batches(trained, precision):
    """TrainLoop.window_batch (the nAveGrad micro-batches of an optimizer step as ONE batch with per-image class counts; bench.py
    --window-fused) against the reference's sequential loop (train_online.py:116-149) on the trained-like net: five different frames, online
    mode.  Same products, other summation order (the weight gradients sum five frames' pixels in one pass instead of five; a batch of five may
    take other tiles / K splits than a batch of one): the summed loss to 1e-6 relative (measured 4e-8 .. 3e-7); the accumulated gradients
    as one vector within 5e-4 and every tensor within 1e-3 -- measured 2e-6 .. 5e-6 per tensor in fp32x3 and, on the exact kernels, either
    the same or 1.4e-4 .. 2.0e-4 across the trunk when ONE ReLU / arg-max decision falls the other way in one of the two runs."""
    from osvos_pytorch_amd.train_common import TrainLoop, make_sgd
    wts, frames, _ = trained
    n_ave = 5
    dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(m).cuda()) for x, m in frames[:n_ave]]

    def run(fused):
        net = tf.build(wts, precision)
        loop = TrainLoop(net, make_sgd(net, "online", lr=0.0), mode="online", n_ave_grad=n_ave)      # lr 0: the step leaves the gradients readable
        keep = {}
        hook = loop.opt.step
        loop.opt.step = lambda: keep.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}) or hook()
        if fused:
            total, stepped = loop.window_batch(torch.cat([d[0] for d in dev]).requires_grad_(), torch.cat([d[1] for d in dev]))
            total = float(total)
        else:
            total = 0.0
            for x, m in dev:
                l, stepped = loop.micro_batch(x.clone().requires_grad_(), m)
                total += float(l)
        assert stepped and loop.steps == 1
        return total, keep
    l_seq, g_seq = run(False)
    l_win, g_win = run(True)
    assert abs(l_win - l_seq) <= 1e-6 * abs(l_seq), (l_win, l_seq)
    assert g_seq.keys() == g_win.keys() and len(g_seq) >= 35
    allw = sorted(((float((g_win[k].double() - g_seq[k].double()).norm() / (g_seq[k].double().norm() + 1e-300)), k) for k in g_seq), reverse=True)
    print("   window-fused worst:", [(k, "%.1e" % e) for e, k in allw[:8]])
    worst = allw[0]
    print("window-fused vs sequential (%s): summed loss rel %.1e, worst gradient rel-L2 %.2e (%s)" % (precision, abs(l_win - l_seq) / abs(l_seq), worst[0], worst[1]))
    num = sum(float((g_win[k].double() - g_seq[k].double()).norm() ** 2) for k in g_seq)
    den = sum(float(g_seq[k].double().norm() ** 2) for k in g_seq)
    assert (num / den) ** 0.5 <= 5e-4 and worst[0] <= 1e-3, (worst, (num / den) ** 0.5)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "fp32x3b2", "fp32h2", "fp32x3h2"])
def test_twenty_optimizer_steps_track_the_float64_trajectory(trained, precision):
    """The online loop for 20 optimizer steps (nAveGrad 2 = 40 micro-batches over four frames) at 60x107 from the trained-like weights,
    through the product's TrainLoop / FusedSGD, against the same loop on the float64 oracle: drift, momentum and weight re-pack bugs show up
    late, the two-step test at 854x480 cannot see them.  Per-step losses within 1e-4 relative, the parameter change of the 20 steps per
    tensor within 3e-3 relative L2 (+ the fp32 resolution of the update)."""
    from oracle import synth, torch_ref
    from osvos_pytorch_amd.train_common import TrainLoop, make_sgd
    wts, _, _ = trained
    n_ave, steps, lr = 2, 20, 5e-8
    frames = [synth.trainable_frame(1, 60, 107, seed=tf.RECIPE["frame_seed"] + 200 + k) for k in range(4)]
    if "traj20" not in _ORACLE_CACHE:      # the float64 trajectory is the same for every precision: computed once per session (15-25 s of host time)
        p = torch_ref.as_leaf_params(wts, dtype=torch.float64)
        opt = torch.optim.SGD(torch_ref.sgd_groups(p, lr=lr, mode="online"), lr=lr, momentum=0.9)
        ref_losses = []
        for it in range(steps * n_ave):
            x, m = frames[it % 4]
            loss, _ = torch_ref.train_loss(p, torch.from_numpy(x).double(), torch.from_numpy(m).double(), mode="online")
            ref_losses.append(float(loss))
            (loss / n_ave).backward()
            if it % n_ave == n_ave - 1:
                opt.step()
                opt.zero_grad()
        _ORACLE_CACHE["traj20"] = (ref_losses, {k: v.detach().clone() for k, v in p.items()})
    ref_losses, p = _ORACLE_CACHE["traj20"]
    net = tf.build(wts, precision)
    loop = TrainLoop(net, make_sgd(net, "online", lr=lr), mode="online", n_ave_grad=n_ave)
    losses = []
    for it in range(steps * n_ave):
        x, m = frames[it % 4]
        l, _ = loop.micro_batch(torch.from_numpy(x).cuda().requires_grad_(), torch.from_numpy(m).cuda())
        losses.append(float(l))
    assert loop.steps == steps
    rel = np.abs(np.array(losses) - np.array(ref_losses)) / np.abs(np.array(ref_losses))
    moved = abs(ref_losses[-1] - ref_losses[1]) / abs(ref_losses[1])
    print("20-step trajectory (%s): max loss rel diff %.2e (first window %.2e); the loss itself moved by %.1f %%" % (precision, rel.max(), rel[:n_ave].max(), 100 * moved))
    assert rel[:n_ave].max() <= 1e-5 and rel.max() <= 2e-4, rel
    sd = net.state_dict()
    for k, v0 in wts.items():
        if k.startswith("upscale") or k.startswith("score_dsn"):
            assert torch.equal(sd[k].cpu(), torch.from_numpy(np.asarray(v0))), k
            continue
        v0t = torch.from_numpy(np.asarray(v0))
        delta = (sd[k].cpu().double() - v0t.double())
        ref_delta = p[k].detach() - v0t.double()
        e = float((delta - ref_delta).norm() / (ref_delta.norm() + 1e-300))
        floor = float(v0t.abs().max()) * 2 ** -23 * np.sqrt(delta.numel()) * np.sqrt(steps) / (float(ref_delta.norm()) + 1e-300)
        assert e <= 3e-3 + 2 * floor, (k, e, floor)
