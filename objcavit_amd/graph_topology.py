"""What a captured forward LOOKS like as a graph: nodes, edges, and the one shape of fork the product allows.

Why this exists: three shapes of a captured forward with side streams replayed 3.5 - 6 ms slower
per step at EVERY batch size, one of them crashed ``hipGraphLaunch`` once and a nested third branch crashed
``hipStreamEndCapture`` (profiles/r04_skip_overlap.txt, profiles/r04_head_overlap.txt), while "one fork, one chain, one join" is
a gain.  Until round 5 the product steered around the bad shapes by convention.  This module reads the captured hipGraph back from
the runtime (``hipGraphGetNodes`` / ``hipGraphGetEdges`` / ``hipGraphNodeGetType``), and ``check`` states the convention as a
property of the DAG, so that ``GraphedGraphBins`` can refuse -- at capture, before anything is replayed -- a forward whose streams
were forked in a shape the runtime is known to replay pathologically.

The property ("a chain of diamonds"): walking from the graph's first node, the graph is a single chain; where a node has two
successors (a FORK) both branches are simple chains (every node one predecessor, one successor) that meet again in one node with
exactly two predecessors (the JOIN), and the walk continues from there.  Equivalently: a side chain depends on the main chain
exactly ONCE (its first node) and the main chain on it exactly ONCE (the join); at most two branches are ever open; forks do not
nest.  Several roots (a side stream that starts at the very top of the capture, before the main stream has a node) and several
leaves are read as a fork at a virtual first node / a join at a virtual last node.

``check`` is pure Python on (node count, edge list): it runs -- and is tested -- without a GPU.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
              8: "ext_sem_signal", 9: "ext_sem_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol",
              13: "memcpy_to_symbol"}


@dataclass
class Topology:
    n: int
    edges: List[Tuple[int, int]]                       # (from, to): ``to`` depends on ``from``; node indices in hipGraphGetNodes order
    types: List[str] = field(default_factory=list)     # NODE_TYPES names, "" when not read
    names: List[str] = field(default_factory=list)     # kernel names, "" for other nodes / when not resolvable

    def degrees(self) -> Tuple[List[int], List[int]]:
        indeg, outdeg = [0] * self.n, [0] * self.n
        for a, b in self.edges:
            outdeg[a] += 1
            indeg[b] += 1
        return indeg, outdeg

    def label(self, i: int) -> str:
        t = self.types[i] if i < len(self.types) and self.types[i] else "node"
        nm = self.names[i] if i < len(self.names) and self.names[i] else ""
        return f"#{i} {t}" + (f" {nm[:60]}" if nm else "")

    def summary(self) -> Dict[str, object]:
        indeg, outdeg = self.degrees()
        kinds: Dict[str, int] = {}
        for t in self.types:
            kinds[t or "?"] = kinds.get(t or "?", 0) + 1
        return {"nodes": self.n, "edges": len(self.edges), "roots": sum(1 for d in indeg if d == 0),
                "leaves": sum(1 for d in outdeg if d == 0), "forks": sum(1 for d in outdeg if d > 1),
                "joins": sum(1 for d in indeg if d > 1), "max_out": max(outdeg, default=0), "max_in": max(indeg, default=0),
                "node_types": kinds}


def check(topo: Topology) -> List[str]:
    """The violations of the chain-of-diamonds property (module docstring); [] = the shape the product allows."""
    n = topo.n
    if n == 0:
        return []
    src, snk = n, n + 1                                  # virtual first / last node
    succ: List[List[int]] = [[] for _ in range(n + 2)]
    pred: List[List[int]] = [[] for _ in range(n + 2)]
    seen = set()
    bad: List[str] = []
    for a, b in topo.edges:
        if not (0 <= a < n and 0 <= b < n) or a == b:
            bad.append(f"edge ({a}, {b}) is not an edge between two different nodes")
            continue
        if (a, b) in seen:
            continue                                     # a duplicate edge is one dependency
        seen.add((a, b))
        succ[a].append(b)
        pred[b].append(a)
    if bad:
        return bad
    # Transitive reduction first: a capture holds edges that add no dependency -- a side STREAM used for two forks in sequence
    # keeps its stream order (last launch of the first side chain -> first launch of the second) although the second chain
    # already waits for a main-chain launch behind the first join (tools/graph_shapes.py 'three_diamonds': 188 edges for 184
    # nodes).  Such an edge is implied by a longer path and is dropped; the edges of the slow shapes (a side chain's SECOND
    # dependency on the main chain) are implied by nothing and stay.
    indeg = [len(pred[i]) for i in range(n)]
    order = [i for i in range(n) if indeg[i] == 0]
    for v in order:
        for w in succ[v]:
            indeg[w] -= 1
            if indeg[w] == 0:
                order.append(w)
    if len(order) != n:
        return ["the graph has a cycle"]
    reach = [0] * n                                       # bit w of reach[v]: w is reachable from v
    for v in reversed(order):
        r = 0
        for w in succ[v]:
            r |= (1 << w) | reach[w]
        reach[v] = r
    for a in range(n):
        if len(succ[a]) > 1:
            keep = [b for b in succ[a] if not any(c != b and (reach[c] >> b) & 1 for c in succ[a])]
            for b in succ[a]:
                if b not in keep:
                    pred[b].remove(a)
            succ[a] = keep
    for i in range(n):
        if not pred[i]:
            succ[src].append(i)
            pred[i].append(src)
        if not succ[i]:
            succ[i].append(snk)
            pred[snk].append(i)

    def lab(i: int) -> str:
        return "the start of the graph" if i == src else "the end of the graph" if i == snk else topo.label(i)

    visited = [False] * (n + 2)
    cur = src
    visited[src] = True
    guard = 0
    while cur != snk:
        guard += 1
        if guard > 4 * (n + 2):
            return bad + ["the graph has a cycle"]
        out = succ[cur]
        if len(out) == 1:
            nxt = out[0]
            if len(pred[nxt]) != 1:
                bad.append(f"{lab(nxt)} has {len(pred[nxt])} predecessors but only the main chain ({lab(cur)}) is open in front of "
                           "it: a join without a fork")
                break
            visited[nxt] = True
            cur = nxt
            continue
        if len(out) != 2:
            bad.append(f"{lab(cur)} forks into {len(out)} branches: more than two parallel branches")
            break
        ends = []
        for first in out:                                # each branch: a simple chain up to the join
            node, prev, ok = first, cur, True
            while True:
                if len(pred[node]) >= 2:
                    break                                # the join (checked below)
                visited[node] = True
                if len(succ[node]) != 1:
                    bad.append(f"{lab(node)}, inside a branch forked at {lab(cur)}, has {len(succ[node])} successors: a fork "
                               "inside an open fork (nested / third branch)")
                    ok = False
                    break
                prev, node = node, succ[node][0]
            if not ok:
                break
            ends.append((node, prev))
        if len(ends) != 2:
            break
        (j0, p0), (j1, p1) = ends
        if j0 != j1:
            a, b = (j0, j1)
            bad.append(f"the branches forked at {lab(cur)} do not meet in one node: one ends in {lab(a)}, the other in {lab(b)} "
                       "(a side chain with a second edge from or to the main chain)")
            break
        if len(pred[j0]) != 2:
            extra = [p for p in pred[j0] if p not in (p0, p1)]
            bad.append(f"{lab(j0)} joins {len(pred[j0])} branches ({', '.join(lab(e) for e in extra)} besides the two forked at "
                       f"{lab(cur)}): a side chain with a second incoming edge")
            break
        visited[j0] = True
        cur = j0
    if not bad:
        missed = [i for i in range(n) if not visited[i]]
        if missed:
            bad.append(f"{len(missed)} node(s) are not on the chain of diamonds, e.g. {lab(missed[0])}")
    return bad


# ---------------------------------------------------------------------------
# reading a captured graph back from the HIP runtime (GPU box only)
# ---------------------------------------------------------------------------
_HIP = None


class _Dim3(ctypes.Structure):
    _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]


class _KernelNodeParams(ctypes.Structure):            # hipKernelNodeParams (hip_runtime_api.h)
    _fields_ = [("blockDim", _Dim3), ("extra", ctypes.c_void_p), ("func", ctypes.c_void_p), ("gridDim", _Dim3),
                ("kernelParams", ctypes.c_void_p), ("sharedMemBytes", ctypes.c_uint)]


def _hip():
    global _HIP
    if _HIP is None:
        h = ctypes.CDLL("libamdhip64.so")
        vp, szp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)
        h.hipGraphGetNodes.argtypes = [vp, vp, szp]
        h.hipGraphGetEdges.argtypes = [vp, vp, vp, szp]
        h.hipGraphNodeGetType.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        h.hipGraphKernelNodeGetParams.argtypes = [vp, ctypes.POINTER(_KernelNodeParams)]
        h.hipKernelNameRefByPtr.argtypes = [vp, vp]
        h.hipKernelNameRefByPtr.restype = ctypes.c_char_p
        h.hipGraphDebugDotPrint.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint]
        for f in (h.hipGraphGetNodes, h.hipGraphGetEdges, h.hipGraphNodeGetType, h.hipGraphKernelNodeGetParams, h.hipGraphDebugDotPrint):
            f.restype = ctypes.c_int
        _HIP = h
    return _HIP


def read(raw_graph: int, kernel_names: bool = True) -> Optional[Topology]:
    """The topology of a captured hipGraph_t (``torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()``); None when the runtime
    cannot be asked."""
    try:
        h = _hip()
        g = ctypes.c_void_p(int(raw_graph))
        n = ctypes.c_size_t(0)
        if h.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * max(1, n.value))()
        if n.value and h.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
            return None
        index = {int(nodes[i] or 0): i for i in range(n.value)}
        m = ctypes.c_size_t(0)
        if h.hipGraphGetEdges(g, None, None, ctypes.byref(m)) != 0:
            return None
        fr, to = (ctypes.c_void_p * max(1, m.value))(), (ctypes.c_void_p * max(1, m.value))()
        if m.value and h.hipGraphGetEdges(g, fr, to, ctypes.byref(m)) != 0:
            return None
        edges = [(index[int(fr[i] or 0)], index[int(to[i] or 0)]) for i in range(m.value)]
        types, names = [], []
        for i in range(n.value):
            t = ctypes.c_int(-1)
            h.hipGraphNodeGetType(nodes[i], ctypes.byref(t))
            types.append(NODE_TYPES.get(t.value, f"type{t.value}"))
            nm = ""
            if kernel_names and t.value == 0:
                kp = _KernelNodeParams()
                if h.hipGraphKernelNodeGetParams(nodes[i], ctypes.byref(kp)) == 0 and kp.func:
                    ref = h.hipKernelNameRefByPtr(kp.func, None)
                    nm = ref.decode("utf-8", "replace") if ref else ""
            names.append(nm)
        return Topology(n.value, edges, types, names)
    except (OSError, AttributeError, KeyError, TypeError, ValueError):
        return None


def dot_print(raw_graph: int, path: str, flags: int = 0) -> bool:
    """``hipGraphDebugDotPrint`` of a captured graph into ``path``."""
    try:
        return _hip().hipGraphDebugDotPrint(ctypes.c_void_p(int(raw_graph)), path.encode(), flags) == 0
    except (OSError, AttributeError):
        return False


def describe(topo: Topology, around: int = 2) -> str:
    """Text listing of the fork and join nodes with their neighbours (for profiles/ records)."""
    indeg, outdeg = topo.degrees()
    succ: Dict[int, List[int]] = {}
    pred: Dict[int, List[int]] = {}
    for a, b in topo.edges:
        succ.setdefault(a, []).append(b)
        pred.setdefault(b, []).append(a)
    lines = [f"{topo.summary()}"]
    for i in range(topo.n):
        if outdeg[i] > 1:
            lines.append(f"  FORK {topo.label(i)} -> " + " | ".join(topo.label(s) for s in succ[i]))
        if indeg[i] > 1:
            lines.append(f"  JOIN {topo.label(i)} <- " + " | ".join(topo.label(s) for s in pred[i]))
        if indeg[i] == 0:
            lines.append(f"  ROOT {topo.label(i)}")
        if outdeg[i] == 0:
            lines.append(f"  LEAF {topo.label(i)}")
        if topo.types and topo.types[i] not in ("kernel", ""):
            lines.append(f"  NON-KERNEL {topo.label(i)} (in {indeg[i]}, out {outdeg[i]})")
    return "\n".join(lines)
