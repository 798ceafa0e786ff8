#!/usr/bin/env python
"""Drop-in for the reference's generate.py (node-classification datasets): load a pre-trained checkpoint, embed every
node of a graph as (f(q) + f(k)) / 2 over its two RWR views with the eval-mode encoder, save
``<model_folder>/<dataset>.npy`` (generate.py:56-125).  The graph comes from ``--edgelist`` (the reference's
``data/<name>/<name>.edgelist`` format, gcc/datasets/data_util.py:61-110) or ``--graph-npz`` (row_ptr/col_idx);
everything runs on the GPU (sampler, positional embedding, encoder).

Extra flags (not in the reference): --edgelist / --nodelabel / --graph-npz / --graphs-npz / --tudataset / --edge-multiplicity /
--batch-size.  Graph-classification datasets (entire_graph=True, generate.py:75-82) come as ``--graphs-npz``: node_off
[G+1], row_ptr [N+1] (per-graph offsets restarting at 0 are rebuilt from node_off), col_idx (local ids)."""
import argparse
import os

import numpy as np
import torch


def main(args_test):
    from gcc_amd import ingest
    from gcc_amd.datasets import GraphClassificationDataset, NodeClassificationDataset
    from gcc_amd.encoder import GraphEncoder
    from gcc_amd.generate import test_moco
    from gcc_amd.posemb import DevicePosEmb

    if os.path.isfile(args_test.load_path):
        print("=> loading checkpoint '{}'".format(args_test.load_path))
        checkpoint = torch.load(args_test.load_path, map_location="cpu", weights_only=False)
        print("=> loaded successfully '{}' (epoch {})".format(args_test.load_path, checkpoint["epoch"]))
    else:
        raise SystemExit("=> no checkpoint found at '{}'".format(args_test.load_path))
    args = checkpoint["opt"]
    assert torch.cuda.is_available(), "the device pipeline needs a GPU"
    args.gpu = 0 if args_test.gpu is None else args_test.gpu
    print("Use GPU: {} for generation".format(args.gpu))
    args.device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(args.device)

    graphs = None
    if args_test.tudataset:
        graphs = ingest.read_tudataset(args_test.tudataset, args_test.dataset)["graphs"]
        graph, mult = None, max(args_test.edge_multiplicity, 1)
    elif args_test.graphs_npz:
        z = np.load(args_test.graphs_npz)
        no, rp, ci = z["node_off"].astype(np.int64), z["row_ptr"].astype(np.int64), z["col_idx"].astype(np.int64)
        graphs = [(rp[no[i]:no[i + 1] + 1] - rp[no[i]], ci[rp[no[i]]:rp[no[i + 1]]]) for i in range(len(no) - 1)]
        graph, mult = None, max(args_test.edge_multiplicity, 1)
    elif args_test.edgelist:
        d = ingest.read_edgelist(args_test.edgelist, args_test.nodelabel, hindex="hindex" in args_test.dataset)
        graph, mult = (d["row_ptr"], d["col_idx"]), d["edge_multiplicity"]
    elif args_test.graph_npz:
        z = np.load(args_test.graph_npz)
        graph, mult = (z["row_ptr"], z["col_idx"]), args_test.edge_multiplicity
    else:
        raise SystemExit("pass --edgelist data/<name>/<name>.edgelist, --graph-npz, --graphs-npz or --tudataset (dataset files are not bundled)")
    if args_test.edge_multiplicity:
        mult = args_test.edge_multiplicity
    if graphs is not None:
        train_dataset = GraphClassificationDataset(                  # generate.py:75-82
            dataset=args_test.dataset, rw_hops=args.rw_hops, subgraph_size=args.subgraph_size,
            restart_prob=args.restart_prob, positional_embedding_size=args.positional_embedding_size,
            graphs=graphs, edge_multiplicity=mult, batch_size=args_test.batch_size, device=args.device)
        node_cap = train_dataset.node_cap
    else:
        train_dataset = NodeClassificationDataset(                   # generate.py:84-91
            dataset=args_test.dataset, rw_hops=args.rw_hops, subgraph_size=args.subgraph_size,
            restart_prob=args.restart_prob, positional_embedding_size=args.positional_embedding_size,
            graph=graph, edge_multiplicity=mult, batch_size=args_test.batch_size, run_seed=getattr(args, "seed", 0),
            device=args.device)
        node_cap = train_dataset.sampler.node_cap
    model = GraphEncoder(                                            # generate.py:102-118
        positional_embedding_size=args.positional_embedding_size, max_node_freq=args.max_node_freq,
        max_edge_freq=args.max_edge_freq, max_degree=args.max_degree, freq_embedding_size=args.freq_embedding_size,
        degree_embedding_size=args.degree_embedding_size, output_dim=args.hidden_size, node_hidden_dim=args.hidden_size,
        edge_hidden_dim=args.hidden_size, num_layers=args.num_layer, num_step_set2set=args.set2set_iter,
        num_layer_set2set=args.set2set_lstm_layer, gnn_model=args.model, norm=args.norm, degree_input=True)
    model = model.to(args.device)
    model.load_state_dict(checkpoint["model"])
    del checkpoint
    posemb = DevicePosEmb(args_test.batch_size, node_cap, args.positional_embedding_size,
                          device=args.device, seed=getattr(args, "seed", 0), max_views=2, num_buffers=2)
    emb = test_moco(train_dataset, model, posemb, args)
    if graphs is None:
        train_dataset.sampler.check_status()
    posemb.check_status()
    os.makedirs(args.model_folder, exist_ok=True)
    out = os.path.join(args.model_folder, args_test.dataset)
    np.save(out, emb.numpy())
    print("saved {}.npy {}".format(out, tuple(emb.shape)))


if __name__ == "__main__":
    parser = argparse.ArgumentParser("argument for training")
    # fmt: off
    parser.add_argument("--load-path", type=str, help="path to load model")
    parser.add_argument("--dataset", type=str, default="dgl")
    parser.add_argument("--gpu", default=None, type=int, help="GPU id to use.")
    # ---- not in the reference: where the graph comes from
    parser.add_argument("--edgelist", type=str, default=None, help="<name>.edgelist of the reference's data folder")
    parser.add_argument("--nodelabel", type=str, default=None, help="<name>.nodelabel (only read to validate the node set)")
    parser.add_argument("--graph-npz", type=str, default=None, help="npz with row_ptr/col_idx of the simple symmetric graph")
    parser.add_argument("--graphs-npz", type=str, default=None, help="npz with node_off/row_ptr/col_idx of a list of small graphs (graph classification)")
    parser.add_argument("--tudataset", type=str, default=None, help="folder with the raw TU files <NAME>_A.txt, <NAME>_graph_indicator.txt, <NAME>_graph_labels.txt of --dataset (imdb-binary, imdb-multi, rdt-b, rdt-5k, collab)")
    parser.add_argument("--edge-multiplicity", type=int, default=0, help="copies of every edge in the reference's DGL graph (edge lists: detected; npz: default 2)")
    parser.add_argument("--batch-size", type=int, default=256)
    # fmt: on
    a = parser.parse_args()
    if a.graph_npz and not a.edge_multiplicity:
        a.edge_multiplicity = 2
    main(a)
