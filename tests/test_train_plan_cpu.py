"""CPU tests of the data-parallel bookkeeping of the parent loop (train_parent.py / osvos_pytorch_amd.train_common):
the per-epoch plan, the world-size rule, and -- over gloo, world size 2 -- the REAL TrainLoop + make_sgd parameter groups on the
reference's module tree, against a single-process run of the same global micro-batch stream (VERDICT r01 item 4 / ADVICE r01)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_epoch_plan_partitions_every_epoch_and_balances_every_step():
    from osvos_pytorch_amd.train_common import epoch_plan
    n_items, n_ave = 2079, 10                  # DAVIS-2016 train: 2079 frames (3^3 * 7 * 11), nAveGrad 10: nothing divides evenly
    for world in (1, 2, 5, 10):
        per_rank_g = [[] for _ in range(world)]
        for epoch in range(3):
            plans = [epoch_plan(n_items, epoch, n_ave, r, world, seed=4) for r in range(world)]
            idx = sorted(i for p in plans for i, _ in p)
            assert idx == list(range(n_items))                     # every frame exactly once per epoch, across ranks
            ref = epoch_plan(n_items, epoch, n_ave, 0, 1, seed=4)     # the single-process order of the same epoch
            merged = sorted(((g, i) for p in plans for i, g in p))
            assert [(g, i) for i, g in ref] == merged                # same permutation on every rank
            for r, p in enumerate(plans):
                per_rank_g[r] += [g for _, g in p]
        # every optimizer step (n_ave consecutive global iterations) holds n_ave / world micro-batches of every rank -> every rank
        # enters the same number of all-reduces, also across the epoch boundaries (3 * 2079 = 623 full steps + 7 left over)
        for r in range(world):
            steps = np.bincount(np.array(per_rank_g[r]) // n_ave)
            assert (steps[:-1] == n_ave // world).all(), (world, r)
    # different epochs / seeds give different orders; shuffle=False is the identity
    assert epoch_plan(50, 0, 5, seed=1) != epoch_plan(50, 1, 5, seed=1) != epoch_plan(50, 1, 5, seed=2)
    assert [i for i, _ in epoch_plan(7, 3, 7, shuffle=False)] == list(range(7))


def test_world_size_must_divide_n_ave_grad():
    from osvos_pytorch_amd.train_common import check_world_divides, epoch_plan
    assert check_world_divides(10, 5) == 2 and check_world_divides(16, 8) == 2 and check_world_divides(10, 1) == 10
    for world in (3, 4, 8):
        with pytest.raises(ValueError, match="not a multiple of the world size"):
            check_world_divides(10, world)
        with pytest.raises(ValueError):
            epoch_plan(100, 0, 10, 0, world)


def _cpu_net(seed):
    """The reference-shaped module tree (same parameter names / groups as the product) with the forward swapped for the CPU oracle:
    the product's forward is HIP-only; the bookkeeping under test is TrainLoop + make_sgd + GradientAllReducer."""
    import networks.vgg_osvos as vo
    from oracle import synth, torch_ref
    sys.stdout, keep = open(os.devnull, "w"), sys.stdout
    try:
        net = vo.OSVOS(pretrained=0)
    finally:
        sys.stdout.close()
        sys.stdout = keep
    wts = synth.calibrate_heads(synth.make_weights(seed), synth.torch_forward_fn(), synth.make_frame(1, 24, 32, seed=3))
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in wts.items()})
    net.forward = lambda x: torch_ref.forward(dict(net.named_parameters()), x)
    return net


def _run_stream(rank, world, n_ave, epochs, n_items):
    from oracle import synth, torch_ref
    from osvos_pytorch_amd.parallel import GradientAllReducer
    from osvos_pytorch_amd.train_common import TrainLoop, check_world_divides, epoch_plan, make_sgd
    net = _cpu_net(seed=1)
    opt = make_sgd(net, 'parent', lr=1e-6, fused=False)          # the reference's 10 parameter groups (lr raised so steps are visible)
    red = GradientAllReducer(net, average=False) if world > 1 else None
    if red is not None:
        red.broadcast_parameters(0)
    loop = TrainLoop(net, opt, mode='parent', n_ave_grad=n_ave, n_epochs=4, reducer=red, local_ave=check_world_divides(n_ave, world),
                     loss_fn=torch_ref.cbce_loss)
    frames = [(torch.from_numpy(synth.make_frame(1, 24, 32, seed=50 + i)), torch.from_numpy(synth.make_mask(1, 24, 32, seed=50 + i)))
              for i in range(n_items)]
    for epoch in range(epochs):
        for idx, _ in epoch_plan(n_items, epoch, n_ave, rank, world, seed=7):
            loop.micro_batch(frames[idx][0], frames[idx][1], epoch=epoch)
    return net, loop


def _worker(rank, world, port, out, n_ave, epochs, n_items):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    net, loop = _run_stream(rank, world, n_ave, epochs, n_items)
    torch.save({"steps": loop.steps, "sd": {k: v.detach().clone() for k, v in net.state_dict().items()}}, out + ".%d" % rank)
    dist.destroy_process_group()


def test_two_rank_parent_loop_equals_single_process_stream(tmp_path):
    """7 frames, nAveGrad 4, 2 epochs: 14 global iterations = 3 optimizer steps + 2 left over, one step straddling the epoch boundary.
    Two ranks (round-robin over the stream, one all-reduce per step, flat arena re-zeroed with reducer.zero_grads) must land on the
    single-process parameters; both ranks must take the same number of steps and hold identical weights."""
    sys.path.insert(0, REPO)
    n_ave, epochs, n_items = 4, 2, 7
    out = str(tmp_path / "dp")
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out, n_ave, epochs, n_items), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    torch.set_num_threads(4)
    net, loop = _run_stream(0, 1, n_ave, epochs, n_items)
    assert r0["steps"] == r1["steps"] == loop.steps == (epochs * n_items) // n_ave == 3
    sd = net.state_dict()
    for k in sd:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k                       # ranks stay in lock step
        torch.testing.assert_close(r0["sd"][k], sd[k], rtol=2e-5, atol=1e-7, msg=k)
    init = _cpu_net(seed=1).state_dict()
    changed = [k for k in sd if not torch.equal(sd[k], init[k])]
    assert any(k.startswith("stages.") for k in changed) and "fuse.weight" in changed      # the steps did move the weights
    assert not any(k.startswith("upscale") for k in changed)                              # lr 0 groups stay put



def test_davis_frame_lists_follow_the_reference_and_prefetcher_needs_cuda(tmp_path):
    """DavisFrames restates the file lists of reference dataloaders/davis_2016.py:36-63 (sorted frames per sequence, first-frame-only
    training list and single annotation for a named sequence) and hands out what cv2.imread would: uint8 BGR + uint8 label."""
    from PIL import Image
    from osvos_pytorch_amd.davis_io import ArrayFrames, DavisFrames, DevicePrefetcher
    root = str(tmp_path)
    rng = np.random.RandomState(0)
    for seq, nf in (("bear", 3), ("swan", 2)):
        os.makedirs(os.path.join(root, "JPEGImages/480p", seq))
        os.makedirs(os.path.join(root, "Annotations/480p", seq))
        for f in range(nf):
            rgb = rng.randint(0, 255, (12, 20, 3)).astype(np.uint8)
            Image.fromarray(rgb).save(os.path.join(root, "JPEGImages/480p", seq, "%05d.png" % f))      # png: lossless, exact check
            Image.fromarray(((rng.rand(12, 20) > 0.5) * 255).astype(np.uint8)).save(os.path.join(root, "Annotations/480p", seq, "%05d.png" % f))
    with open(os.path.join(root, "train_seqs.txt"), "w") as f:
        f.write("swan\nbear\n")
    with open(os.path.join(root, "val_seqs.txt"), "w") as f:
        f.write("bear\n")
    tr = DavisFrames(True, root)
    assert [os.path.basename(os.path.dirname(p)) for p in tr.img_list] == ["swan", "swan", "bear", "bear", "bear"] and len(tr) == 5
    assert [os.path.basename(p) for p in tr.labels] == ["00000.png", "00001.png", "00000.png", "00001.png", "00002.png"]
    img, lab = tr[2]
    with Image.open(os.path.join(root, tr.img_list[2])) as im:
        assert np.array_equal(img, np.asarray(im)[:, :, ::-1]) and img.dtype == np.uint8 and img.shape == (12, 20, 3)      # BGR like cv2.imread
    assert lab.dtype == np.uint8 and set(np.unique(lab)) <= {0, 255}
    assert len(DavisFrames(False, root)) == 3
    one_tr, one_te = DavisFrames(True, root, seq_name="bear"), DavisFrames(False, root, seq_name="bear")
    assert len(one_tr) == 1 and len(one_te) == 3 and one_te.labels == [os.path.join("Annotations/480p/", "bear", "00000.png"), None, None]
    assert one_te[1][1] is None and one_te.fname(2) == os.path.join("bear", "00002")
    with pytest.raises(RuntimeError, match="CUDA"):
        DevicePrefetcher(ArrayFrames([(img, lab)]), [0], "cpu")


def test_step_schedule_closes_every_epoch_once_and_in_collective_order():
    """The round-2 deadlock, as arithmetic: with 2079 frames and nAveGrad 10 no epoch ends on a step boundary.  Simulate the collective
    sequence every rank would issue (gradient all-reduce per complete window, statistics all-reduce per closed epoch) for W in {1,2,5,10}
    and for a run whose tail window is partial: all ranks must issue the SAME sequence."""
    from osvos_pytorch_amd.train_common import StepSchedule, check_world_divides, epoch_plan
    # (nAveGrad 8 / 16: the plans of a 4- and an 8-GPU node, whose world size does not divide the reference's 10 -- SURVEY 8e)
    for (n_items, n_ave, epochs, first) in [(2079, 10, 3, 0), (7, 4, 2, 0), (13, 10, 3, 1), (20, 10, 2, 0), (2079, 8, 2, 0), (2079, 16, 2, 0), (37, 16, 3, 1)]:
        sched = StepSchedule(n_items, n_ave, first, epochs)
        assert sched.total_steps == ((epochs - first) * n_items) // n_ave
        seqs = []
        for world in [w for w in (1, 2, 4, 5, 8, 10, 16) if n_ave % w == 0]:
            local_ave = check_world_divides(n_ave, world)
            for rank in range(world):
                seq, ave, steps, pending = [], 0, 0, []
                for epoch in range(first, epochs):
                    pending.append(epoch)
                    for _ in epoch_plan(n_items, epoch, n_ave, rank, world, seed=3):
                        ave += 1
                        if ave % local_ave == 0 and steps < sched.total_steps:      # TrainLoop.micro_batch's rule
                            ave, steps = 0, steps + 1
                            seq.append(("grad", steps - 1))
                            for e in sched.closed_by(steps, pending):
                                pending.remove(e)
                                seq.append(("stats", e))
                for e in list(pending):
                    seq.append(("stats", e))
                assert steps == sched.total_steps, (n_items, world, rank)
                seqs.append(seq)
        assert all(s == seqs[0] for s in seqs), (n_items, n_ave)
        assert [e for kind, e in seqs[0] if kind == "stats"] == list(range(first, epochs))          # every epoch once, in order


def _main_worker(rank, world, port, out, argv):
    sys.path.insert(0, REPO)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    os.environ["OSVOS_SAVE_ROOT"] = os.path.dirname(out)
    torch.set_num_threads(2)
    import train_parent
    from oracle import torch_ref
    net, loop = train_parent.main(argv, build_net=lambda: _cpu_net(seed=1), loss_fn=torch_ref.cbce_loss)
    torch.save({"steps": loop.steps, "sd": {k: v.detach().clone() for k, v in net.state_dict().items()}, "validation": dict(loop.validation)},
               out + ".%d" % rank)
    if dist.is_initialized():
        dist.destroy_process_group()


def test_train_parent_main_two_ranks_over_epoch_boundaries(tmp_path):
    """train_parent.main() itself on two gloo ranks (network forward and loss swapped for the CPU oracle): 7 frames, nAveGrad 4, 3 epochs =
    21 iterations = 5 complete windows + 1 left over; no epoch ends on a step boundary, and rank 0 owns the whole tail.  Both ranks must
    come back (round 2: the per-epoch statistics all-reduce paired with the other rank's gradient all-reduce), take the same number
    of steps, hold identical weights, and match the single-process run of the same script."""
    sys.path.insert(0, REPO)
    argv = ["--synthetic", "7", "--epochs", "3", "--n-ave-grad", "4", "--height", "24", "--width", "32", "--seed", "7", "--lr", "1e-6",
            "--test-interval", "1"]       # validation after EVERY epoch: sharded over the ranks, its sums exchanged at the next closing step
    out = str(tmp_path / "main")
    port = 31700 + (os.getpid() % 2000)
    mp.spawn(_main_worker, args=(2, port, out, argv), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    mp.spawn(_main_worker, args=(1, port + 1, out + "_single", argv), nprocs=1, join=True)
    single = torch.load(out + "_single.0")
    assert r0["steps"] == r1["steps"] == single["steps"] == 5
    for k in single["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
        torch.testing.assert_close(r0["sd"][k], single["sd"][k], rtol=2e-5, atol=1e-7, msg=k)
    init = _cpu_net(seed=1).state_dict()
    assert any(not torch.equal(single["sd"][k], init[k]) for k in single["sd"] if k.startswith("stages."))
    # the sharded validation pass: every epoch reported once, over the whole validation set, the same numbers on both ranks and (weights
    # agree to 2e-5) the single process's numbers
    assert sorted(r0["validation"]) == sorted(r1["validation"]) == sorted(single["validation"]) == [0, 1, 2]
    for e in range(3):
        assert r0["validation"][e][1] == single["validation"][e][1] == 2
        np.testing.assert_allclose(r0["validation"][e][0], r1["validation"][e][0], rtol=0, atol=0)
        np.testing.assert_allclose(r0["validation"][e][0], single["validation"][e][0], rtol=1e-3)


@pytest.mark.parametrize("world", [1, 2])
def test_exact_resume_with_optimizer_state_continues_bit_for_bit(tmp_path, world):
    """--save-optimizer (round 4): train_parent.main() over 4 epochs of 7 frames with nAveGrad 4 writes, right after the optimizer
    step that closes epoch 1 -- the window {12..15} straddles the boundary between epochs 1 and 2 -- the network, the SGD momentum buffers and the
    position in the global iteration stream (16: two iterations into epoch 2).  A second run resumed from that bundle (--resume-epoch 2) must
    end on the SAME BITS as the uninterrupted one -- single process, two gloo ranks, and a bundle written by ONE process resumed on TWO ranks
    (equal up to the all-reduce's summation order); a reference-style resume (network only: momentum and the open window lost) must not."""
    sys.path.insert(0, REPO)
    base = ["--synthetic", "7", "--epochs", "4", "--n-ave-grad", "4", "--height", "24", "--width", "32", "--seed", "7", "--lr", "1e-6", "--snapshot", "1"]
    port = 33100 + (os.getpid() % 1500) + 10 * world
    full = str(tmp_path / "full")
    mp.spawn(_main_worker, args=(world, port, full, base + ["--save-optimizer"]), nprocs=world, join=True)
    a = [torch.load(full + ".%d" % r) for r in range(world)]
    ck = torch.load(str(tmp_path / "parent_epoch-1.optim.pth"), weights_only=False)
    assert ck["next_iteration"] == 16 and ck["world"] == world and ck["optimizer"]["state"] and len(ck["net"]) == 52
    assert sorted(ck["partial_stats"]) == [2] and ck["partial_stats"][2][1] == 2          # iterations 14, 15 of epoch 2 already ran
    import shutil
    keep = str(tmp_path / "bundle_epoch1.pth")
    shutil.copy(str(tmp_path / "parent_epoch-1.optim.pth"), keep)
    res = str(tmp_path / "resumed")
    mp.spawn(_main_worker, args=(world, port + 1, res, base + ["--resume-epoch", "2"]), nprocs=world, join=True)      # (the bundle is used because it exists)
    b = [torch.load(res + ".%d" % r) for r in range(world)]
    assert a[0]["steps"] == 7 and b[0]["steps"] == 7 - 4          # 28 iterations = 7 windows; 4 of them closed before the bundle
    for r in range(world):
        for k in a[0]["sd"]:
            assert torch.equal(a[r]["sd"][k], b[r]["sd"][k]), (world, r, k)
    # the reference's resume (no optimizer file read): a different trajectory
    ref = str(tmp_path / "refstyle")
    mp.spawn(_main_worker, args=(world, port + 2, ref, base + ["--resume-epoch", "2", "--no-resume-optimizer"]), nprocs=world, join=True)
    c = torch.load(ref + ".0")
    assert any(not torch.equal(a[0]["sd"][k], c["sd"][k]) for k in a[0]["sd"] if k.startswith("stages."))
    if world == 1:      # a single-process bundle continues on two ranks: same windows, same frames, the gradient sum in another order
        shutil.copy(keep, str(tmp_path / "parent_epoch-1.optim.pth"))
        two = str(tmp_path / "resumed_on_two")
        mp.spawn(_main_worker, args=(2, port + 3, two, base + ["--save-optimizer", "--resume-epoch", "2"]), nprocs=2, join=True)
        d = torch.load(two + ".0")
        assert d["steps"] == 3
        for k in a[0]["sd"]:
            torch.testing.assert_close(d["sd"][k], a[0]["sd"][k], rtol=2e-5, atol=1e-7, msg=k)


def test_a_stale_optimizer_bundle_does_not_outlive_a_new_snapshot(tmp_path):
    """run 1 writes parent_epoch-1.pth + parent_epoch-1.optim.pth; run 2 in the same directory (another seed, NO
    --save-optimizer) rewrites parent_epoch-1.pth.  The old bundle must be gone, so that `--resume-epoch 2` continues from run 2's snapshot the
    way the reference does (train_parent.py:59-65 loads the .pth only) instead of silently taking run 1's weights, momentum and position."""
    sys.path.insert(0, REPO)
    base = ["--synthetic", "7", "--epochs", "2", "--n-ave-grad", "4", "--height", "24", "--width", "32", "--lr", "1e-6", "--snapshot", "1"]
    port = 37300 + (os.getpid() % 1500)
    mp.spawn(_main_worker, args=(1, port, str(tmp_path / "run1"), base + ["--seed", "7", "--save-optimizer"]), nprocs=1, join=True)
    assert os.path.exists(str(tmp_path / "parent_epoch-1.optim.pth"))
    mp.spawn(_main_worker, args=(1, port + 1, str(tmp_path / "run2"), base + ["--seed", "8"]), nprocs=1, join=True)
    assert os.path.exists(str(tmp_path / "parent_epoch-1.pth")) and not os.path.exists(str(tmp_path / "parent_epoch-1.optim.pth"))
    snap = torch.load(str(tmp_path / "parent_epoch-1.pth"))
    run2 = torch.load(str(tmp_path / "run2.0"))
    for k in snap:
        assert torch.equal(snap[k], run2["sd"][k]), k
    # a run WITH the flag keeps (rewrites) its own bundle
    mp.spawn(_main_worker, args=(1, port + 2, str(tmp_path / "run3"), base + ["--seed", "9", "--save-optimizer"]), nprocs=1, join=True)
    ck = torch.load(str(tmp_path / "parent_epoch-1.optim.pth"), weights_only=False)
    run3 = torch.load(str(tmp_path / "run3.0"))
    assert all(torch.equal(ck["net"][k], run3["sd"][k]) for k in ck["net"])


def test_optimizer_bundle_with_a_window_that_takes_one_iteration_from_the_next_epoch(tmp_path):
    """7 frames, nAveGrad 4, snapshot 1 -- the window {32..35} that closes epoch 4 holds exactly ONE iteration (35) of
    epoch 5, and on two ranks only rank 1 runs it.  save_bundle used to size its statistics all-reduce from the rank's OWN counters: rank 0
    skipped it and its next gradient all-reduce paired with rank 1's statistics exchange (gloo aborted with a collective mismatch).  Both
    ranks must come back, agree bit for bit, and the bundle must carry that one iteration's statistics."""
    sys.path.insert(0, REPO)
    argv = ["--synthetic", "7", "--epochs", "6", "--n-ave-grad", "4", "--height", "24", "--width", "32", "--seed", "7", "--lr", "1e-6",
            "--snapshot", "1", "--save-optimizer"]
    out = str(tmp_path / "asym")
    port = 34700 + (os.getpid() % 1500)
    mp.spawn(_main_worker, args=(2, port, out, argv), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["steps"] == r1["steps"] == 10          # 42 iterations = 10 windows + 2 left over
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    ck = torch.load(str(tmp_path / "parent_epoch-4.optim.pth"), weights_only=False)
    assert ck["next_iteration"] == 36 and sorted(ck["partial_stats"]) == [5] and ck["partial_stats"][5][1] == 1
    ck1 = torch.load(str(tmp_path / "parent_epoch-1.optim.pth"), weights_only=False)
    assert ck1["next_iteration"] == 16 and ck1["partial_stats"][2][1] == 2


def test_train_parent_main_four_ranks_with_n_ave_grad_8(tmp_path):
    """The plan of a 4-GPU node (nAveGrad 8: W = 4 does not divide the reference's 10) through train_parent.main() on four gloo ranks, so
    that the first multi-GPU run with W in {4, 8} is not also the first run of that plan code: 11 frames, 3 epochs = 33 iterations = 4
    complete windows + 1 left over; equal weights on all ranks and the single-process trajectory up to summation order."""
    sys.path.insert(0, REPO)
    argv = ["--synthetic", "11", "--epochs", "3", "--n-ave-grad", "8", "--height", "24", "--width", "32", "--seed", "5", "--lr", "1e-6",
            "--test-interval", "2"]
    out = str(tmp_path / "four")
    port = 36100 + (os.getpid() % 1500)
    mp.spawn(_main_worker, args=(4, port, out, argv), nprocs=4, join=True)
    rs = [torch.load(out + ".%d" % r) for r in range(4)]
    mp.spawn(_main_worker, args=(1, port + 1, out + "_single", argv), nprocs=1, join=True)
    single = torch.load(out + "_single.0")
    assert [r["steps"] for r in rs] == [4] * 4 and single["steps"] == 4
    for k in single["sd"]:
        for r in rs[1:]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), k
        torch.testing.assert_close(rs[0]["sd"][k], single["sd"][k], rtol=2e-5, atol=1e-7, msg=k)
    assert sorted(rs[0]["validation"]) == sorted(single["validation"]) == [1]
    np.testing.assert_allclose(rs[0]["validation"][1][0], rs[3]["validation"][1][0], rtol=0, atol=0)


@pytest.mark.parametrize("n_ave", [8, 16])
def test_train_parent_main_eight_ranks(tmp_path, n_ave):
    """The plan of an 8-GPU node through train_parent.main() on EIGHT gloo ranks (W <= 4 end to end until now): nAveGrad 8
    (one micro-batch per rank and optimizer step) and 16 (two); 19 frames, 2 epochs = 38 iterations = 4 (2) complete windows + a partial one;
    every rank ends on the same bits, and on the single-process trajectory up to the summation order of the gradient all-reduce
    (reference: train_parent.py:46,163-172 -- nAveGrad micro-batches of batch 1 per optimizer step)."""
    sys.path.insert(0, REPO)
    argv = ["--synthetic", "19", "--epochs", "2", "--n-ave-grad", str(n_ave), "--height", "24", "--width", "32", "--seed", "3", "--lr", "1e-6"]
    out = str(tmp_path / "eight")
    port = 38100 + (os.getpid() % 1500) + n_ave
    mp.spawn(_main_worker, args=(8, port, out, argv), nprocs=8, join=True)
    rs = [torch.load(out + ".%d" % r) for r in range(8)]
    mp.spawn(_main_worker, args=(1, port + 1, out + "_single", argv), nprocs=1, join=True)
    single = torch.load(out + "_single.0")
    want = 38 // n_ave
    assert [r["steps"] for r in rs] == [want] * 8 and single["steps"] == want
    for k in single["sd"]:
        for r in rs[1:]:
            assert torch.equal(rs[0]["sd"][k], r["sd"][k]), k
        torch.testing.assert_close(rs[0]["sd"][k], single["sd"][k], rtol=2e-5, atol=1e-7, msg=k)


def _split_root_worker(rank, world, port, roots, argv, out):
    sys.path.insert(0, REPO)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    os.environ["OSVOS_SAVE_ROOT"] = roots[rank]
    torch.set_num_threads(2)
    import train_parent
    from oracle import torch_ref
    try:
        train_parent.main(argv, build_net=lambda: _cpu_net(seed=1), loss_fn=torch_ref.cbce_loss)
        msg = "ran"
    except SystemExit as e:
        msg = "exit: %s" % (e,)
    with open(out + ".%d" % rank, "w") as f:
        f.write(msg)
    if dist.is_initialized():
        dist.destroy_process_group()


def test_resume_bundle_visible_to_one_rank_only_is_refused_on_every_rank(tmp_path):
    """only rank 0 writes the optimizer bundle; without a shared filesystem rank 1 would fall back to the reference-style resume
    (another start iteration, another collective sequence).  The ranks exchange "I see it" first and all of them stop with the same message."""
    sys.path.insert(0, REPO)
    base = ["--synthetic", "7", "--epochs", "3", "--n-ave-grad", "4", "--height", "24", "--width", "32", "--seed", "7", "--lr", "1e-6", "--snapshot", "1"]
    roots = [str(tmp_path / "node0"), str(tmp_path / "node1")]
    for r in roots:
        os.makedirs(r)
    port = 37300 + (os.getpid() % 1500)
    mp.spawn(_main_worker, args=(1, port, os.path.join(roots[0], "w"), base + ["--save-optimizer"]), nprocs=1, join=True)
    assert os.path.exists(os.path.join(roots[0], "parent_epoch-1.optim.pth"))
    import shutil
    shutil.copy(os.path.join(roots[0], "parent_epoch-1.pth"), os.path.join(roots[1], "parent_epoch-1.pth"))      # the network snapshot is on both nodes
    out = str(tmp_path / "split")
    mp.spawn(_split_root_worker, args=(2, port + 1, roots, base + ["--resume-epoch", "2"], out), nprocs=2, join=True)
    msgs = [open(out + ".%d" % r).read() for r in range(2)]
    assert all(m.startswith("exit: exact resume:") and "1 of 2 ranks" in m for m in msgs), msgs
