"""Graph files of the reference's downstream datasets -> int32 CSR.

``<name>.edgelist`` / ``<name>.nodelabel`` as read by ``Edgelist._preprocess`` (gcc/datasets/data_util.py:61-110):
one "u v" pair per line, node ids re-indexed in order of first appearance; "node label" per line.  The reference then
inserts every pair in both directions (data_util.py:84-85) and its dataset class inserts both directions AGAIN
(graph_dataset.py:301-302), so its DGL graph is a multigraph in which every undirected edge of the file exists
``2 x (times the pair is listed, in either order)`` times per direction.  The HIP path keeps a simple CSR plus ONE
uniform multiplicity (gcc_gin_pass.edge_multiplicity); files whose pairs repeat a non-uniform number of times, and self
loops, are rejected rather than approximated.

``read_tudataset``: the raw TU Dortmund collection layout (``<NAME>_A.txt``, ``<NAME>_graph_indicator.txt``,
``<NAME>_graph_labels.txt``) that DGL's ``TUDataset`` downloads for ``create_graph_classification_dataset``
(data_util.py:47-58: imdb-binary, imdb-multi, rdt-b, rdt-5k, collab) -> the list of small graphs
``GraphClassificationDataset(graphs=...)`` takes, plus the graph labels."""
from __future__ import annotations

import numpy as np


def csr_from_pairs(pairs: np.ndarray, num_nodes: int):
    """pairs int64 [m, 2] (undirected, any order, repeats allowed) -> (row_ptr, col_idx, multiplicity)."""
    if (pairs[:, 0] == pairs[:, 1]).any():
        raise ValueError("self loops are not supported by the sampler contract (x2dgl.py:41-42 removes them)")
    lo, hi = pairs.min(axis=1), pairs.max(axis=1)
    key, counts = np.unique(lo * num_nodes + hi, return_counts=True)
    if counts.min() != counts.max():
        raise ValueError("pairs repeat a non-uniform number of times: a general multigraph is not supported")
    u, v = key // num_nodes, key % num_nodes
    src = np.concatenate([u, v])
    dst = np.concatenate([v, u])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    if (np.diff(row_ptr) == 0).any():
        raise ValueError("isolated nodes (every node of an edge list has an edge; check the input)")
    return row_ptr.astype(np.int32), dst.astype(np.int32), int(counts[0])


def read_edgelist(edgelist_path: str, nodelabel_path: str = None, hindex: bool = False):
    """-> dict(row_ptr, col_idx, edge_multiplicity, node2id, y).  ``edge_multiplicity`` is the number of copies of
    every edge in the graph the reference's NodeClassificationDataset builds (2 x listings)."""
    node2id, pairs = {}, []
    with open(edgelist_path) as f:
        for line in f:
            if not line.strip():
                continue
            x, y = (int(t) for t in line.split()[:2])
            for n in (x, y):
                if n not in node2id:
                    node2id[n] = len(node2id)
            pairs.append((node2id[x], node2id[y]))
    num_nodes = len(node2id)
    row_ptr, col_idx, listed = csr_from_pairs(np.asarray(pairs, dtype=np.int64), num_nodes)
    out = dict(row_ptr=row_ptr, col_idx=col_idx, edge_multiplicity=2 * listed, node2id=node2id, y=None)
    if nodelabel_path is not None:
        nodes, labels, label2id = [], [], {}
        with open(nodelabel_path) as f:
            for line in f:
                if not line.strip():
                    continue
                x, label = (int(t) for t in line.split()[:2])
                if label not in label2id:
                    label2id[label] = len(label2id)
                nodes.append(node2id[x])
                labels.append(label if hindex else label2id[label])
        if hindex:                                   # data_util.py:104-106
            median = np.median(labels)
            labels = [int(l > median) for l in labels]
        y = np.zeros((num_nodes, max(len(label2id), 1)), dtype=np.float32)
        y[nodes, labels] = 1
        out["y"] = y
    return out


TU_NAMES = {"imdb-binary": "IMDB-BINARY", "imdb-multi": "IMDB-MULTI", "rdt-b": "REDDIT-BINARY",
            "rdt-5k": "REDDIT-MULTI-5K", "collab": "COLLAB"}                      # data_util.py:48-54


def read_tudataset(folder: str, name: str):
    """-> dict(graphs=[(row_ptr, col_idx), ...], graph_labels int64 [G], num_labels).  ``name`` is the reference's
    dataset name (``imdb-binary`` ...) or the TU name itself.  Nodes of a graph keep their file order (DGL builds each
    graph as the subgraph of its ascending node ids); labels are re-indexed 0..C-1 in ascending order of the file's
    values, as DGL's TUDataset does.  The adjacency must be symmetric without self loops or repeated entries."""
    import os

    tu = TU_NAMES.get(name, name)
    base = os.path.join(folder, tu + "_")
    edges = np.loadtxt(base + "A.txt", delimiter=",", dtype=np.int64, ndmin=2) - 1
    indicator = np.loadtxt(base + "graph_indicator.txt", dtype=np.int64, ndmin=1)
    labels = np.loadtxt(base + "graph_labels.txt", dtype=np.int64, ndmin=1)
    num_nodes = len(indicator)
    if (np.diff(indicator) < 0).any():
        raise ValueError("graph_indicator must be non-decreasing (nodes of a graph are contiguous in the TU format)")
    gids, first = np.unique(indicator, return_index=True)
    if len(gids) != len(labels):
        raise ValueError(f"{len(gids)} graphs in graph_indicator but {len(labels)} graph labels")
    src, dst = edges[:, 0], edges[:, 1]
    if len(src) and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= num_nodes):
        raise ValueError("node id out of range in A.txt")
    if (src == dst).any():
        raise ValueError("self loops are not supported by the sampler contract (x2dgl.py:41-42 removes them)")
    if (indicator[src] != indicator[dst]).any():
        raise ValueError("an edge connects two different graphs")
    key = src * num_nodes + dst
    if len(np.unique(key)) != len(key):
        raise ValueError("repeated adjacency entries: a general multigraph is not supported")
    if not np.array_equal(np.sort(key), np.sort(dst * num_nodes + src)):
        raise ValueError("the adjacency is not symmetric")
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    bounds = np.append(first, num_nodes)
    graphs = []
    for g in range(len(gids)):
        lo, hi = int(bounds[g]), int(bounds[g + 1])
        rp = row_ptr[lo:hi + 1] - row_ptr[lo]
        graphs.append((rp.astype(np.int32), (dst[row_ptr[lo]:row_ptr[hi]] - lo).astype(np.int32)))
    values = np.unique(labels)
    return dict(graphs=graphs, graph_labels=np.searchsorted(values, labels).astype(np.int64), num_labels=len(values))


# ---------------------------------------------------------------------------------------------------------------
# DGL graph files (``data/small.bin``): what ``dgl.data.utils.save_graphs`` / ``load_graphs`` / ``load_labels`` of
# dgl 0.4.x write and read -- graph_dataset.py:26-29 (worker_init_fn), graph_dataset.py:58-60 (graph_sizes label),
# x2dgl.py:119-131 (the converter that produces the pre-training corpus).
#
# FORMAT STATUS: DGL-recalled, parity UNPINNED.  DGL is neither vendored in /root/reference nor installable here, and
# the reference ships no .bin file, so the container layout below is a restatement from memory of dgl 0.4.3's
# src/graph/graph_serialize.cc and the DLPack tensor blobs of src/runtime/ndarray.cc; the reader checks every magic
# number and size it relies on and raises on the first mismatch instead of guessing.  Layout (little endian):
#
#   0     u64 magic 0xDD2E4FF046B4A13F | u64 version (1) | u64 graph type (1 = immutable) | zero padding to 4096
#   4096  u64 num_graph
#         vector<u64> graph_offsets   (dmlc vector = u64 count + elements: absolute file offset of each graph)
#         vector<u64> num_nodes, vector<u64> num_edges
#         vector<pair<string, tensor>> labels          (string = u64 length + bytes; here: "graph_sizes")
#   per graph, at its offset: tensor indptr[n+1], tensor indices[nnz], tensor edge_ids[nnz]  (the IN-CSR, int64),
#         vector<pair<string, tensor>> node data, vector<pair<string, tensor>> edge data
#   tensor = u64 magic 0xDD5E40F096B4A13F | u64 reserved | i32 device_type (1 = CPU) | i32 device_id | i32 ndim |
#            u8 dtype code (0 int, 1 uint, 2 float) | u8 bits | u16 lanes | i64 shape[ndim] | i64 nbytes | data
_DGL_MAGIC = 0xDD2E4FF046B4A13F
_NDARRAY_MAGIC = 0xDD5E40F096B4A13F
_DT_CODES = {0: "i", 1: "u", 2: "f"}


class _Reader:
    def __init__(self, buf, name):
        self.b, self.o, self.name = memoryview(buf), 0, name

    def take(self, fmt):
        import struct

        size = struct.calcsize("<" + fmt)
        if self.o + size > len(self.b):
            raise ValueError(f"{self.name}: truncated at byte {self.o}")
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += size
        return v if len(v) > 1 else v[0]

    def u64_vector(self):
        n = self.take("Q")
        if n > (len(self.b) - self.o) // 8:
            raise ValueError(f"{self.name}: vector of {n} entries does not fit the file (offset {self.o})")
        v = np.frombuffer(self.b, dtype="<u8", count=n, offset=self.o).copy()
        self.o += 8 * n
        return v

    def string(self):
        n = self.take("Q")
        if n > 4096:
            raise ValueError(f"{self.name}: implausible string length {n} at byte {self.o}")
        s = bytes(self.b[self.o:self.o + n]).decode()
        self.o += n
        return s

    def tensor(self):
        magic, _reserved = self.take("QQ")
        if magic != _NDARRAY_MAGIC:
            raise ValueError(f"{self.name}: tensor magic {magic:#x} at byte {self.o - 16} (expected {_NDARRAY_MAGIC:#x})")
        dev_type, _dev_id, ndim = self.take("iii")
        code, bits, lanes = self.take("BBH")
        if dev_type != 1 or lanes != 1 or code not in _DT_CODES or ndim < 0 or ndim > 8:
            raise ValueError(f"{self.name}: unsupported tensor header (device {dev_type}, dtype {code}/{bits}/{lanes}, ndim {ndim})")
        shape = [self.take("q") for _ in range(ndim)]
        nbytes = self.take("q")
        count = int(np.prod(shape)) if ndim else 1
        if nbytes != count * bits // 8 or self.o + nbytes > len(self.b):
            raise ValueError(f"{self.name}: tensor of shape {shape} with {nbytes} bytes at byte {self.o}")
        a = np.frombuffer(self.b, dtype=f"<{_DT_CODES[code]}{bits // 8}", count=count, offset=self.o).reshape(shape).copy()
        self.o += nbytes
        return a

    def named_tensors(self):
        return {self.string(): self.tensor() for _ in range(self.take("Q"))}


def _dgl_header(r):
    magic, version, gtype = r.take("QQQ")
    if magic != _DGL_MAGIC:
        raise ValueError(f"{r.name}: not a DGL graph file (magic {magic:#x}, expected {_DGL_MAGIC:#x})")
    if version != 1 or gtype != 1:
        raise ValueError(f"{r.name}: DGL graph file version {version} / graph type {gtype}; only the dgl 0.4.x immutable-graph "
                         "container (version 1, type 1) is understood")
    r.o = 4096
    num = r.take("Q")
    offsets, nodes, edges = r.u64_vector(), r.u64_vector(), r.u64_vector()
    if not (len(offsets) == len(nodes) == len(edges) == num):
        raise ValueError(f"{r.name}: graph table sizes disagree ({num}, {len(offsets)}, {len(nodes)}, {len(edges)})")
    return offsets, nodes, edges, r.named_tensors()


def read_dgl_labels(path: str) -> dict:
    """``dgl.data.utils.load_labels(path)`` (graph_dataset.py:58-60): the label dict only, without touching the graphs."""
    with open(path, "rb") as f:
        r = _Reader(f.read(), path)
    return _dgl_header(r)[3]


def read_dgl_graphs(path: str, idx_list=None, validate: bool = True):
    """``dgl.data.utils.load_graphs(path, idx_list)`` (graph_dataset.py:26-29) -> ([(row_ptr, col_idx) int32, ...], labels).
    The stored structure is the in-CSR; the sampler contract (x2dgl.py:39-62: symmetric, no self loops, no duplicates,
    no isolated nodes -- so in-CSR == out-CSR) is checked per graph unless ``validate=False``."""
    from .graphgen import check_contract

    with open(path, "rb") as f:
        r = _Reader(f.read(), path)
    offsets, nodes, edges, labels = _dgl_header(r)
    graphs = []
    for i in (range(len(offsets)) if idx_list is None else idx_list):
        r.o = int(offsets[i])
        indptr, indices, _edge_ids = r.tensor(), r.tensor(), r.tensor()
        r.named_tensors()                                   # node data (x2dgl.py:122 clears it)
        r.named_tensors()                                   # edge data
        if len(indptr) != nodes[i] + 1 or len(indices) != edges[i] or indptr[-1] != edges[i]:
            raise ValueError(f"{path}: graph {i}: CSR of {len(indptr) - 1} nodes / {len(indices)} edges, table says {nodes[i]} / {edges[i]}")
        if edges[i] >= 2 ** 31 or nodes[i] >= 2 ** 31:
            raise ValueError(f"{path}: graph {i} does not fit int32 ids")
        rp, ci = indptr.astype(np.int32), indices.astype(np.int32)
        if validate:
            # DGL does not promise sorted rows (an in-CSR materialised from COO input keeps insertion order and carries a
            # permutation in edge_ids); the sampler does.  One vectorised pass over ALL rows: a descent is only allowed
            # where a new row starts.
            if len(ci) > 1:
                descent = np.flatnonzero(np.diff(ci.astype(np.int64)) <= 0) + 1       # positions whose predecessor is >= them
                starts = np.zeros(len(ci) + 1, dtype=bool)
                starts[rp[:-1]] = True
                if not np.all(starts[descent]):
                    order = np.lexsort((ci, np.repeat(np.arange(len(rp) - 1), np.diff(rp))))
                    ci = ci[order]
            check_contract(rp, ci)
        graphs.append((rp, ci))
    if idx_list is None and "graph_sizes" in labels and len(labels["graph_sizes"]) != len(graphs):
        raise ValueError(f"{path}: label graph_sizes has {len(labels['graph_sizes'])} entries for {len(graphs)} graphs "
                         "(x2dgl.py:120,129-131 writes one node count per graph)")
    return graphs, labels


def write_dgl_graphs(path: str, graphs, labels: dict = None):
    """The inverse of :func:`read_dgl_graphs` (``save_graphs(filename, g_list, labels)``, x2dgl.py:129-131) for symmetric
    CSR graphs.  Same format status: DGL-recalled, unpinned -- it exists so that corpora can be prepared without DGL and
    so that the tests can round-trip the reader."""
    import struct

    def tensor(a):
        a = np.ascontiguousarray(a)
        code = {"i": 0, "u": 1, "f": 2}[a.dtype.kind]
        head = struct.pack("<QQiiiBBH", _NDARRAY_MAGIC, 0, 1, 0, a.ndim, code, a.dtype.itemsize * 8, 1)
        return head + struct.pack(f"<{a.ndim}q", *a.shape) + struct.pack("<q", a.nbytes) + a.astype(a.dtype.newbyteorder("<")).tobytes()

    def named(d):
        out = struct.pack("<Q", len(d))
        for k, v in d.items():
            kb = k.encode()
            out += struct.pack("<Q", len(kb)) + kb + tensor(np.asarray(v))
        return out

    def vec(v):
        return struct.pack("<Q", len(v)) + np.asarray(v, dtype="<u8").tobytes()

    blobs = []
    for rp, ci in graphs:
        rp, ci = np.asarray(rp, dtype=np.int64), np.asarray(ci, dtype=np.int64)
        blobs.append(tensor(rp) + tensor(ci) + tensor(np.arange(len(ci), dtype=np.int64)) + named({}) + named({}))
    nodes = [len(rp) - 1 for rp, _ in graphs]
    edges = [len(ci) for _, ci in graphs]
    lab = named(labels or {})
    table = 4096 + 8 + 3 * (8 + 8 * len(graphs)) + len(lab)
    offsets, o = [], table
    for b in blobs:
        offsets.append(o)
        o += len(b)
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", _DGL_MAGIC, 1, 1).ljust(4096, b"\0"))
        f.write(struct.pack("<Q", len(graphs)) + vec(offsets) + vec(nodes) + vec(edges) + lab)
        for b in blobs:
            f.write(b)
