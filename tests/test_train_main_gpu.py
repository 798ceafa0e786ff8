"""The documented entry points end to end on a real MI355X (scripts/pretrain.sh, reference README.md:79-83):
train.py on a DGL graph file (read without DGL) with --moco -> checkpoint -> --resume -> generate.py on that
checkpoint.  Checks the drop-in contract of the files (checkpoint dictionary of train.py:748-786, state_dict keys,
model folder name) and that the fused step actually trains (loss falls on a small corpus)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _corpus(tmp_path):
    from gcc_amd import ingest
    from gcc_amd.graphgen import powerlaw_graph

    gs = [powerlaw_graph(20000, 200000, 0), powerlaw_graph(6000, 50000, 1), powerlaw_graph(2500, 20000, 2)]
    sizes = np.array([len(rp) - 1 for rp, _ in gs], dtype=np.int64)
    path = tmp_path / "small.bin"
    ingest.write_dgl_graphs(str(path), gs, labels={"graph_sizes": sizes})      # x2dgl.py:129-131
    return str(path), gs


def test_pretrain_resume_generate(tmp_path, capsys):
    import generate
    import train

    corpus, gs = _corpus(tmp_path)
    common = ["--exp", "Pretrain", "--model-path", str(tmp_path / "saved"), "--tb-path", str(tmp_path / "tb"),
              "--gpu", "0", "--moco", "--nce-k", "256", "--batch-size", "32", "--num-workers", "4", "--num-copies", "2", "--num-samples", "64",
              "--rw-hops", "64", "--dgl-file", corpus, "--print-freq", "2", "--tb-freq", "4", "--save-freq", "1",
              "--producer-lanes", "2", "--producer-chunk", "2"]
    args = train.parse_option(common + ["--epochs", "2"])
    args.gpu = args.gpu[0]
    loss1 = train.main(args)                                   # 2 epochs x (64 * 4 // 32 = 8) steps
    out = capsys.readouterr().out
    assert "Train: [2][8/8]" in out and np.isfinite(loss1)
    folder = args.model_folder
    assert os.path.basename(folder).startswith("Pretrain_moco_True_dgl_gin_layer_5_lr_0.005_decay_1e-05_bsz_32_hid_64_samples_64_")
    ckpt = torch.load(os.path.join(folder, "current.pth"), map_location="cpu", weights_only=False)
    assert set(ckpt) == {"opt", "model", "contrast", "optimizer", "epoch", "model_ema"} and ckpt["epoch"] == 2
    assert set(ckpt["contrast"]) == {"params", "memory"} and ckpt["contrast"]["memory"].shape == (256, 64)
    assert os.path.isfile(os.path.join(folder, "ckpt_epoch_1.pth"))
    assert any(k.startswith("gnn.ginlayers.0.apply_func.mlp.linears.0") for k in ckpt["model"])
    w_before = ckpt["model"]["gnn.ginlayers.0.apply_func.mlp.linears.0.weight"].clone()

    # --resume: weights / queue / EMA come from the checkpoint, training continues (train.py:487-506,685-702)
    # (4 epochs: the reference's schedule lr * warmup_linear((epoch * n_batch + idx) / (epochs * n_batch)) starts the count
    #  at epoch 1 and is 0 throughout a 1-epoch run)
    args2 = train.parse_option(common + ["--epochs", "4", "--resume", os.path.join(folder, "current.pth")])
    args2.gpu = args2.gpu[0]
    loss2 = train.main(args2)
    out2 = capsys.readouterr().out
    assert "loaded successfully" in out2 and np.isfinite(loss2)
    ckpt2 = torch.load(os.path.join(folder, "current.pth"), map_location="cpu", weights_only=False)
    assert not torch.equal(ckpt2["model"]["gnn.ginlayers.0.apply_func.mlp.linears.0.weight"], w_before)
    assert loss2 < 1.2 * loss1                                   # keeps training from where it was, does not restart

    # generate.py on the checkpoint: every node of a graph gets (f(q) + f(k)) / 2 (generate.py:33-53)
    rp, ci = gs[2]
    npz = tmp_path / "g.npz"
    np.savez(npz, row_ptr=rp, col_idx=ci)
    a = types.SimpleNamespace(load_path=os.path.join(folder, "current.pth"), dataset="toy", gpu=0, edgelist=None,
                              nodelabel=None, graph_npz=str(npz), graphs_npz=None, tudataset=None, edge_multiplicity=2,
                              batch_size=64)
    generate.main(a)
    emb = np.load(os.path.join(folder, "toy.npy"))
    assert emb.shape == (len(rp) - 1, 64) and np.isfinite(emb).all()
    norms = np.linalg.norm(emb, axis=1)
    assert norms.max() <= 1.0 + 1e-4 and norms.min() > 0.05        # mean of two unit vectors


def test_moco_loss_stays_sane_over_an_epoch_and_every_step_is_metered(tmp_path):
    """48 fused steps at bsz 32, K 256: the log lines cover ALL steps (device-side accumulation, one read-back per line),
    the running loss stays finite and near log(257) = 5.55 or below (no blow-up from the flat Adam / EMA / clip path)."""
    import train

    corpus, _ = _corpus(tmp_path)
    args = train.parse_option(["--exp", "T", "--model-path", str(tmp_path / "s"), "--tb-path", str(tmp_path / "t"), "--gpu", "0",
                               "--moco", "--nce-k", "256", "--batch-size", "32", "--num-workers", "1", "--num-copies", "1", "--num-samples", "1536",
                               "--rw-hops", "64", "--dgl-file", corpus, "--epochs", "1", "--print-freq", "8", "--tb-freq", "1000",
                               "--producer-lanes", "2", "--producer-chunk", "2", "--learning_rate", "0.005"])
    args.gpu = args.gpu[0]
    import io
    from contextlib import redirect_stdout

    buf = io.StringIO()
    with redirect_stdout(buf):
        train.main(args)
    vals = [float(l.split("loss ")[1].split(" ")[0]) for l in buf.getvalue().splitlines() if l.startswith("Train:")]
    assert len(vals) == 6 and all(np.isfinite(v) and 0.0 < v < 7.0 for v in vals), vals
    print("loss per 8 steps:", vals)


@pytest.mark.parametrize("flags,tag", [
    ([], "e2e-fused-adam"),                                        # train.py:396-417 through E2ETrainStep
    (["--optimizer", "adagrad"], "e2e-adagrad"),                   # train.py:672-678 through the API path
    (["--moco", "--optimizer", "sgd", "--nce-k", "256"], "moco-sgd"),   # train.py:659-664, momentum 0.9
])
def test_e2e_mode_and_the_other_optimizers_through_train_py(tmp_path, flags, tag):
    """BASELINE configs[0] (E2E: K = bsz - 1, NCESoftmaxLossNS) and --optimizer sgd | adagrad (train.py:658-679) run
    through train.py's main on the multi-graph corpus: finite falling-or-flat loss, checkpoint written with the
    reference's keys, every step metered."""
    import io
    from contextlib import redirect_stdout

    import train

    corpus, _ = _corpus(tmp_path)
    argv = ["--exp", tag, "--model-path", str(tmp_path / "s"), "--tb-path", str(tmp_path / "t"), "--gpu", "0",
            "--batch-size", "32", "--num-workers", "2", "--num-copies", "1", "--num-samples", "256",
            "--rw-hops", "64", "--dgl-file", corpus, "--epochs", "2", "--print-freq", "4", "--tb-freq", "1000",
            "--producer-lanes", "2", "--producer-chunk", "2"]
    if "--nce-k" not in flags:
        argv += ["--nce-k", "31"]                                   # K = batch_size - 1 (README.md:69-75)
    args = train.parse_option(argv + flags)
    args.gpu = args.gpu[0]
    buf = io.StringIO()
    with redirect_stdout(buf):
        loss = train.main(args)
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("Train:")]
    vals = [float(l.split("loss ")[1].split(" ")[0]) for l in lines]
    assert len(vals) == 8 and all(np.isfinite(v) and 0.0 < v < 7.0 for v in vals), (tag, vals)   # 2 epochs x 16 steps / 4
    assert np.isfinite(loss)
    ckpt = torch.load(os.path.join(args.model_folder, "current.pth"), map_location="cpu", weights_only=False)
    want = {"opt", "model", "contrast", "optimizer", "epoch"} | ({"model_ema"} if "--moco" in flags else set())
    assert set(ckpt) == want and ckpt["epoch"] == 2
    if "--moco" not in flags:       # in-batch negatives: around ln(32) = 3.47 at the start (T = 0.07 puts it a little above), falling
        assert vals[0] < 4.5 and min(vals[1:]) < vals[0] and vals[-1] < vals[0] + 0.2, (tag, vals)
