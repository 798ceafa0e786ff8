#!/usr/bin/env python
"""train.py -- drop-in for /root/reference/train.py on the pre-training path.

Same flag surface (all 49 flags of train.py:45-120, same names/types/defaults),
same model_name / folder layout (train.py:133-166), same checkpoint dictionary
({opt, model, contrast, optimizer, epoch[, model_ema]}, train.py:748-786) and the
same per-step semantics (train.py:378-434), executed by the MI355X-native hot
path of gcc_amd.  Out of scope: --finetune / --cv (downstream
supervised loops) and the non-"dgl" evaluation datasets.

The pre-training corpus ./data/small.bin (a DGL graph file, train.py:552) is read
without DGL (gcc_amd/ingest.py).  Extra flags (not in the reference): --dgl-file,
--graph-npz / --synthetic choose another pre-training graph.
Multi-GPU: launch with torch.distributed.run; the seed batch is sharded by rank,
keys are all-gathered before the enqueue, gradients are all-reduced (RCCL).
"""
import argparse
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")    # few hardware queues: see gcc_amd.train_step.BatchProducer
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # RCCL's stream must not share a hardware queue with a producer lane

import numpy as np
import psutil
import torch

from gcc_amd.contrast import MemoryMoCo, NCESoftmaxLoss, NCESoftmaxLossNS, e2e_logits
from gcc_amd.encoder import GraphEncoder
from gcc_amd.misc import AverageMeter, adjust_learning_rate, warmup_linear
from gcc_amd.sampler import LoadBalanceGraphDataset
from gcc_amd.train_step import E2ETrainStep, MoCoTrainStep, clip_grad_norm, flatten_parameters, moment_update, read_meters

GRAPH_CLASSIFICATION_DSETS = ["collab", "imdb-binary", "imdb-multi", "rdt-b", "rdt-5k"]


def parse_option(argv=None):
    # fmt: off
    parser = argparse.ArgumentParser("argument for training")

    parser.add_argument("--print-freq", type=int, default=10, help="print frequency")
    parser.add_argument("--tb-freq", type=int, default=250, help="tb frequency")
    parser.add_argument("--save-freq", type=int, default=1, help="save frequency")
    parser.add_argument("--batch-size", type=int, default=32, help="batch_size")
    parser.add_argument("--num-workers", type=int, default=12, help="num of workers to use")
    parser.add_argument("--num-copies", type=int, default=6, help="num of dataset copies that fit in memory")
    parser.add_argument("--num-samples", type=int, default=2000, help="num of samples per batch per worker")
    parser.add_argument("--epochs", type=int, default=100, help="number of training epochs")

    # optimization
    parser.add_argument("--optimizer", type=str, default='adam', choices=['sgd', 'adam', 'adagrad'], help="optimizer")
    parser.add_argument("--learning_rate", type=float, default=0.005, help="learning rate")
    parser.add_argument("--lr_decay_epochs", type=str, default="120,160,200", help="where to decay lr, can be a list")
    parser.add_argument("--lr_decay_rate", type=float, default=0.0, help="decay rate for learning rate")
    parser.add_argument("--beta1", type=float, default=0.9, help="beta1 for adam")
    parser.add_argument("--beta2", type=float, default=0.999, help="beta2 for Adam")
    parser.add_argument("--weight-decay", type=float, default=1e-5, help="weight decay")
    parser.add_argument("--momentum", type=float, default=0.9, help="momentum")
    parser.add_argument("--clip-norm", type=float, default=1.0, help="clip norm")

    # resume
    parser.add_argument("--resume", default="", type=str, metavar="PATH", help="path to latest checkpoint (default: none)")

    # augmentation setting
    parser.add_argument("--aug", type=str, default="1st", choices=["1st", "2nd", "all"])

    parser.add_argument("--exp", type=str, default="")

    # dataset definition
    parser.add_argument("--dataset", type=str, default="dgl", choices=["dgl", "wikipedia", "blogcatalog", "usa_airport", "brazil_airport", "europe_airport", "cora", "citeseer", "pubmed", "kdd", "icdm", "sigir", "cikm", "sigmod", "icde", "h-index-rand-1", "h-index-top-1", "h-index"] + GRAPH_CLASSIFICATION_DSETS)

    # model definition
    parser.add_argument("--model", type=str, default="gin", choices=["gat", "mpnn", "gin"])
    parser.add_argument("--num-layer", type=int, default=5, help="gnn layers")
    parser.add_argument("--readout", type=str, default="avg", choices=["avg", "set2set"])
    parser.add_argument("--set2set-lstm-layer", type=int, default=3, help="lstm layers for s2s")
    parser.add_argument("--set2set-iter", type=int, default=6, help="s2s iteration")
    parser.add_argument("--norm", action="store_true", default=True, help="apply 2-norm on output feats")

    # loss function
    parser.add_argument("--nce-k", type=int, default=32)
    parser.add_argument("--nce-t", type=float, default=0.07)

    # random walk
    parser.add_argument("--rw-hops", type=int, default=256)
    parser.add_argument("--subgraph-size", type=int, default=128)
    parser.add_argument("--restart-prob", type=float, default=0.8)
    parser.add_argument("--hidden-size", type=int, default=64)
    parser.add_argument("--positional-embedding-size", type=int, default=32)
    parser.add_argument("--max-node-freq", type=int, default=16)
    parser.add_argument("--max-edge-freq", type=int, default=16)
    parser.add_argument("--max-degree", type=int, default=512)
    parser.add_argument("--freq-embedding-size", type=int, default=16)
    parser.add_argument("--degree-embedding-size", type=int, default=16)

    # specify folder
    parser.add_argument("--model-path", type=str, default=None, help="path to save model")
    parser.add_argument("--tb-path", type=str, default=None, help="path to tensorboard")
    parser.add_argument("--load-path", type=str, default=None, help="loading checkpoint at test time")

    # memory setting
    parser.add_argument("--moco", action="store_true", help="using MoCo (otherwise Instance Discrimination)")

    # finetune setting
    parser.add_argument("--finetune", action="store_true")

    parser.add_argument("--alpha", type=float, default=0.999, help="exponential moving average weight")

    # GPU setting
    parser.add_argument("--gpu", default=None, type=int, nargs='+', help="GPU id to use.")

    # cross validation
    parser.add_argument("--seed", type=int, default=0, help="random seed.")
    parser.add_argument("--fold-idx", type=int, default=0, help="random seed.")
    parser.add_argument("--cv", action="store_true")

    # ---- not in the reference: where the pre-training graph comes from
    parser.add_argument("--dgl-file", type=str, default="./data/small.bin", help="DGL graph file of the pre-training corpus (train.py:552 hard-codes this path)")
    parser.add_argument("--graph-npz", type=str, default=None, help="npz with row_ptr/col_idx (instead of data/small.bin)")
    parser.add_argument("--synthetic", type=str, default=None, help="V,E of a synthetic power-law graph, e.g. 1000000,10000000")
    parser.add_argument("--nce-dtype", type=str, default="f32", choices=["f32", "bf16"], help="operands of the MoCo head: f32 = exact (1e-3 parity with the reference), bf16 = matrix-core throughput mode")
    parser.add_argument("--max-steps", type=int, default=0, help="stop after this many steps (0 = full schedule)")
    parser.add_argument("--producer-lanes", type=int, default=2, help="data-pipeline streams (the GPU's command processor serves few queues well)")
    parser.add_argument("--producer-chunk", type=int, default=4, help="steps a lane prepares per turn (2x as many views per eigensolver call, <= 32)")
    # fmt: on

    opt = parser.parse_args(argv)

    iterations = opt.lr_decay_epochs.split(",")
    opt.lr_decay_epochs = list([])
    for it in iterations:
        opt.lr_decay_epochs.append(int(it))

    return opt


def option_update(opt):
    """train.py:133-166, verbatim naming so that checkpoint folders are interchangeable."""
    opt.model_name = "{}_moco_{}_{}_{}_layer_{}_lr_{}_decay_{}_bsz_{}_hid_{}_samples_{}_nce_t_{}_nce_k_{}_rw_hops_{}_restart_prob_{}_aug_{}_ft_{}_deg_{}_pos_{}_momentum_{}".format(
        opt.exp, opt.moco, opt.dataset, opt.model, opt.num_layer, opt.learning_rate, opt.weight_decay,
        opt.batch_size, opt.hidden_size, opt.num_samples, opt.nce_t, opt.nce_k, opt.rw_hops, opt.restart_prob,
        opt.aug, opt.finetune, opt.degree_embedding_size, opt.positional_embedding_size, opt.alpha,
    )
    if opt.load_path is None:
        opt.model_folder = os.path.join(opt.model_path, opt.model_name)
        if not os.path.isdir(opt.model_folder):
            os.makedirs(opt.model_folder, exist_ok=True)
    else:
        opt.model_folder = opt.load_path
    opt.tb_folder = os.path.join(opt.tb_path, opt.model_name)
    if not os.path.isdir(opt.tb_folder):
        os.makedirs(opt.tb_folder, exist_ok=True)
    return opt


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass


def _summary_writer(folder):
    try:
        from torch.utils.tensorboard import SummaryWriter     # train.py:22,706

        return SummaryWriter(folder)
    except Exception:
        return _NullWriter()


def _load_graph(args):
    from gcc_amd.graphgen import powerlaw_graph

    if args.graph_npz:
        z = np.load(args.graph_npz)
        return z["row_ptr"], z["col_idx"]
    if args.synthetic:
        v, e = (int(x) for x in args.synthetic.split(","))
        return powerlaw_graph(v, e, seed=0)
    return None                                           # ./data/small.bin, read by LoadBalanceGraphDataset (train.py:552)


def train_moco(epoch, dataset, trainer, model, model_ema, contrast, criterion, optimizer, sw, opt, posemb):
    """one epoch training for moco -- train.py:350-478"""
    n_batch = dataset.total // opt.batch_size
    batch_time, data_time = AverageMeter(), AverageMeter()
    loss_meter, epoch_loss_meter, prob_meter = AverageMeter(), AverageMeter(), AverageMeter()
    graph_size, gnorm_meter = AverageMeter(), AverageMeter()
    max_num_nodes = max_num_edges = 0
    end = time.time()
    it = None if trainer is not None else iter(dataset)
    # every step's loss / prob / gnorm / graph sizes are accumulated ON THE DEVICE (no host sync) and read back when
    # a log line is due, so the meters cover all steps like the reference's (which synchronises every step, :433).
    # Fused steps do it inside the step, before the batch's ring slot is released (gcc_step_meters).
    dev = next(model.parameters()).device
    acc = torch.zeros(6, dtype=torch.float64, device=dev)      # sums: loss, prob, gnorm, nodes(q+k), steps; [5] unused
    mx = torch.zeros(2, dtype=torch.int32, device=dev)         # max nodes, max edges of a q view
    one = torch.ones(1, dtype=torch.float64, device=dev)
    for idx in range(n_batch):
        global_step = epoch * n_batch + idx
        lr_this_step = opt.learning_rate * warmup_linear(global_step / (opt.epochs * n_batch), 0.1)   # :411-414
        bsz = opt.batch_size
        if trainer is not None:                          # fused step: MoCo (train.py:387-431) or E2E (:396-417)
            trainer.step((epoch - 1) * n_batch + idx, lr_this_step)
        else:                                            # API path (autograd + a torch optimizer): --optimizer sgd / adagrad
            graph_q, graph_k = next(it)
            posemb(graph_q)
            posemb(graph_k)
            data_time.update(time.time() - end)
            feat_q = model(graph_q)
            if opt.moco:                                 # train.py:388-394
                with torch.no_grad():
                    feat_k = model_ema(graph_k)
                out = contrast(feat_q, feat_k)
            else:                                        # train.py:396-401
                feat_k = model(graph_k)
                out = e2e_logits(feat_q, feat_k, opt.nce_t)
            prob = out.prob
            optimizer.zero_grad()
            loss = criterion(out)
            loss.backward()
            grad_norm = clip_grad_norm(list(model.parameters()), opt.clip_norm)
            for param_group in optimizer.param_groups:
                param_group["lr"] = lr_this_step
            optimizer.step()
            if opt.moco:
                moment_update(model, model_ema, opt.alpha)   # train.py:430-431
            B_ = graph_q.batch_size
            nodes_qk = (graph_q.node_off[B_] + graph_k.node_off[B_]).to(torch.float64).reshape(1)
            acc[:5] += torch.cat([loss.detach().reshape(1).double(), prob.detach().reshape(1).double(),
                                  torch.as_tensor(grad_norm, device=dev).detach().reshape(1).double(), nodes_qk, one])
            mx.copy_(torch.maximum(mx, torch.stack([graph_q.node_off[B_], graph_q.edge_off[B_]])))
        want_log = (idx + 1) % opt.print_freq == 0 or (idx + 1) % opt.tb_freq == 0 or idx + 1 == n_batch \
            or (opt.max_steps and (epoch - 1) * n_batch + idx + 1 >= opt.max_steps)
        if want_log:                                     # one read-back per log line
            if trainer is not None:
                a, m = read_meters(trainer)
                trainer.check_status()                   # overflow / refusal flags: never train on for an epoch unseen
            else:
                a, m = acc.tolist(), mx.tolist()
                acc.zero_()
                mx.zero_()
            cnt = max(int(a[4]), 1)
            loss_meter.update(a[0] / cnt, bsz * cnt)
            epoch_loss_meter.update(a[0] / cnt, bsz * cnt)
            prob_meter.update(a[1] / cnt, bsz * cnt)
            graph_size.update(a[3] / cnt / 2.0 / bsz, 2 * bsz * cnt)
            gnorm_meter.update(a[2] / cnt, cnt)
            max_num_nodes = max(max_num_nodes, m[0])
            max_num_edges = max(max_num_edges, m[1])
        batch_time.update(time.time() - end)
        end = time.time()
        if (idx + 1) % opt.print_freq == 0:
            mem = psutil.virtual_memory()
            print("Train: [{0}][{1}/{2}]\t"
                  "BT {batch_time.val:.3f} ({batch_time.avg:.3f})\t"
                  "DT {data_time.val:.3f} ({data_time.avg:.3f})\t"
                  "loss {loss.val:.3f} ({loss.avg:.3f})\t"
                  "prob {prob.val:.3f} ({prob.avg:.3f})\t"
                  "GS {graph_size.val:.3f} ({graph_size.avg:.3f})\t"
                  "mem {mem:.3f}".format(epoch, idx + 1, n_batch, batch_time=batch_time, data_time=data_time,
                                         loss=loss_meter, prob=prob_meter, graph_size=graph_size,
                                         mem=mem.used / 1024 ** 3))
        if (idx + 1) % opt.tb_freq == 0:
            sw.add_scalar("moco_loss", loss_meter.avg, global_step)
            sw.add_scalar("moco_prob", prob_meter.avg, global_step)
            sw.add_scalar("graph_size", graph_size.avg, global_step)
            sw.add_scalar("graph_size/max", max_num_nodes, global_step)
            sw.add_scalar("graph_size/max_edges", max_num_edges, global_step)
            sw.add_scalar("gnorm", gnorm_meter.avg, global_step)
            sw.add_scalar("learning_rate", lr_this_step, global_step)
            loss_meter.reset(); prob_meter.reset(); graph_size.reset(); gnorm_meter.reset()
            max_num_nodes, max_num_edges = 0, 0
        if opt.max_steps and (epoch - 1) * n_batch + idx + 1 >= opt.max_steps:
            break
    return epoch_loss_meter.avg


def main(args):
    np.random.seed(args.seed)                             # train.py:483-486
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    if args.finetune or args.cv or args.dataset != "dgl":
        raise NotImplementedError("--finetune/--cv and the evaluation datasets are outside the accelerated "
                                  "pre-training path (SURVEY.md §2.1 #5, #7)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not args.moco:
        raise NotImplementedError("only the MoCo step is data parallel (seed batch sharded by rank, key all-gather, gradient "
                                  "all-reduce); the E2E / --nce-k 0 path would train N identical replicas: run it on one GPU")
    checkpoint = None
    if args.resume:                                       # train.py:487-506
        if os.path.isfile(args.resume):
            print("=> loading checkpoint '{}'".format(args.resume))
            checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
            pretrain_args = checkpoint["opt"]
            for name in ("fold_idx", "gpu", "finetune", "resume", "cv", "dataset", "epochs", "num_workers",
                         "batch_size", "dgl_file", "graph_npz", "synthetic", "max_steps", "producer_lanes", "producer_chunk"):
                setattr(pretrain_args, name, getattr(args, name))
            args = pretrain_args
        else:
            print("=> no checkpoint found at '{}'".format(args.resume))
    args = option_update(args)
    print(args)
    assert args.gpu is not None and torch.cuda.is_available()     # train.py:509
    print("Use GPU: {} for training".format(args.gpu))
    assert args.positional_embedding_size % 2 == 0
    torch.cuda.set_device(args.gpu)
    dev = torch.device("cuda", args.gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)

    train_dataset = LoadBalanceGraphDataset(                 # train.py:547-556
        rw_hops=args.rw_hops, restart_prob=args.restart_prob,
        positional_embedding_size=args.positional_embedding_size, num_workers=args.num_workers,
        num_samples=args.num_samples, dgl_graphs_file=args.dgl_file, num_copies=args.num_copies,
        graph=_load_graph(args), batch_size=args.batch_size, run_seed=args.seed, device=dev)

    model, model_ema = [
        GraphEncoder(                                      # train.py:601-620
            positional_embedding_size=args.positional_embedding_size, max_node_freq=args.max_node_freq,
            max_edge_freq=args.max_edge_freq, max_degree=args.max_degree,
            freq_embedding_size=args.freq_embedding_size, degree_embedding_size=args.degree_embedding_size,
            output_dim=args.hidden_size, node_hidden_dim=args.hidden_size, edge_hidden_dim=args.hidden_size,
            num_layers=args.num_layer, num_step_set2set=args.set2set_iter, num_layer_set2set=args.set2set_lstm_layer,
            norm=args.norm, gnn_model=args.model, degree_input=True).to(dev)
        for _ in range(2)
    ]
    flatten_parameters(model)
    flatten_parameters(model_ema)
    if args.moco:
        moment_update(model, model_ema, 0)                # copy weights, train.py:623-624
    contrast = MemoryMoCo(args.hidden_size, None, args.nce_k, args.nce_t, use_softmax=True,
                          nce_dtype=getattr(args, "nce_dtype", "f32")).to(dev)                                # :627-629
    criterion = NCESoftmaxLoss() if args.moco else NCESoftmaxLossNS()                                 # :634
    from gcc_amd.posemb import DevicePosEmb
    from gcc_amd.sampler import DeviceRWRSampler

    posemb = None                                         # API path only; the fused steps embed in their producer lanes
    trainer, optimizer = None, None
    wide = model.wide or contrast.wide    # above 64 channels (or a wider input): the any-width kernels (csrc/ginx.hip)
    # --moco with Adam: the fused step at every width (MoCoTrainStep: producer lanes, flat buffers, clip + Adam + EMA as two launches,
    # data parallel; above 64 channels on the any-width kernels, launch by launch).  E2E above 64 channels and SGD / Adagrad: the API
    # path below (single GPU)
    if wide and world > 1 and not (args.moco and args.optimizer == "adam"):
        raise NotImplementedError("--hidden-size above 64 on several GPUs needs --moco with --optimizer adam (the data-parallel step)")
    if args.optimizer == "adam" and (not wide or args.moco):
        # data pipeline: `producer_lanes` streams, each preparing `producer_chunk` steps per turn (sampler calls + one
        # multi-view eigensolver call) -- the role of the reference's --num-workers DataLoader processes
        lanes, depth = [], 2
        for _ in range(args.producer_lanes):
            smp = DeviceRWRSampler(train_dataset.graph, args.batch_size, run_seed=args.seed,
                                   num_buffers=depth * args.producer_chunk, max_steps=args.producer_chunk)
            lanes.append((smp, DevicePosEmb(args.batch_size, smp.node_cap, args.positional_embedding_size, device=dev,
                                            seed=args.seed, num_buffers=depth * args.producer_chunk,
                                            max_views=min(2 * args.producer_chunk, 32))))
        if args.moco:
            trainer = MoCoTrainStep(model, model_ema, contrast, lanes[0][0], lanes[0][1],
                                    learning_rate=args.learning_rate, betas=(args.beta1, args.beta2),
                                    weight_decay=args.weight_decay, clip_norm=args.clip_norm, alpha=args.alpha,
                                    world_size=world, rank=rank, lanes=lanes, depth=depth, chunk=args.producer_chunk)
        else:
            trainer = E2ETrainStep(model, lanes[0][0], lanes[0][1], nce_t=args.nce_t, learning_rate=args.learning_rate,
                                   betas=(args.beta1, args.beta2), weight_decay=args.weight_decay,
                                   clip_norm=args.clip_norm, lanes=lanes, depth=depth, chunk=args.producer_chunk)
        optimizer = trainer.optimizer
        # the loop reads results only behind read_meters() (which joins the step's stream) and the epoch's device
        # synchronisation: no per-step stream hand-offs (gcc_amd/train_step.py: MoCoTrainStep.step)
        trainer.relaxed_streams = True
    else:
        # train.py:658-679: SGD(momentum) / Adagrad through autograd and torch.optim -- the API path of the same kernels; also
        # Adam for models wider than the fused step's 64 channels
        if world > 1:
            raise NotImplementedError("--optimizer sgd/adagrad runs the single-GPU API path; the data-parallel step is fused Adam")
        posemb = DevicePosEmb(args.batch_size, train_dataset.node_cap, args.positional_embedding_size,
                              device=dev, seed=args.seed)
        if args.optimizer == "adam":                      # (--hidden-size above 64) train.py:667-672
            optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate, betas=(args.beta1, args.beta2),
                                         weight_decay=args.weight_decay)
        elif args.optimizer == "sgd":
            optimizer = torch.optim.SGD(model.parameters(), lr=args.learning_rate, momentum=args.momentum,
                                        weight_decay=args.weight_decay)
        else:
            optimizer = torch.optim.Adagrad(model.parameters(), lr=args.learning_rate, lr_decay=args.lr_decay_rate,
                                            weight_decay=args.weight_decay)
        model.train()
        if args.moco:                                     # train.py:357-365
            model_ema.eval()
            for mod in model_ema.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.train()

    args.start_epoch = 1
    if checkpoint is not None:                            # train.py:685-702 (optimizer state deliberately not restored)
        model.load_state_dict(checkpoint["model"])
        contrast.load_state_dict(checkpoint["contrast"])
        if args.moco:
            model_ema.load_state_dict(checkpoint["model_ema"])
        print("=> loaded successfully '{}' (epoch {})".format(args.resume, checkpoint["epoch"]))
        del checkpoint
        torch.cuda.empty_cache()

    sw = _summary_writer(args.tb_folder) if rank == 0 else _NullWriter()
    for epoch in range(args.start_epoch, args.epochs + 1):
        adjust_learning_rate(epoch, args, optimizer)
        print("==> training...")
        time1 = time.time()
        loss = train_moco(epoch, train_dataset, trainer, model, model_ema, contrast, criterion, optimizer, sw, args,
                          posemb)
        torch.cuda.synchronize()
        print("epoch {}, total time {:.2f}".format(epoch, time.time() - time1))
        # overflow / refusal flags of everything that produced this epoch's batches, BEFORE the checkpoint is written:
        # "raises, never truncates"
        if trainer is not None:
            trainer.check_status()
        else:
            train_dataset.sampler.check_status()
            posemb.check_status()
        if rank == 0:                                     # train.py:748-786
            state = {"opt": args, "model": model.state_dict(), "contrast": contrast.state_dict(),
                     "optimizer": optimizer.state_dict(), "epoch": epoch}
            if args.moco:
                state["model_ema"] = model_ema.state_dict()
            if epoch % args.save_freq == 0:
                torch.save(state, os.path.join(args.model_folder, "ckpt_epoch_{epoch}.pth".format(epoch=epoch)))
            torch.save(state, os.path.join(args.model_folder, "current.pth"))
            del state
        if args.max_steps:
            break
    if world > 1:
        torch.distributed.destroy_process_group()
    return loss


if __name__ == "__main__":
    args = parse_option()
    if args.gpu is None:
        args.gpu = [int(os.environ.get("LOCAL_RANK", "0"))]
    assert args.gpu is not None and torch.cuda.is_available()
    args.gpu = args.gpu[0]                                 # train.py:817
    main(args)
