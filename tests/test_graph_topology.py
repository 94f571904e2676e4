"""The chain-of-diamonds rule of objcavit_amd/graph_topology.py: pure DAG logic, no GPU.  The shapes are the
ones of profiles/r04_skip_overlap.txt / r04_head_overlap.txt: what the product captures, and the three that replayed 3.5 - 6 ms slow
or crashed."""
import pytest

from objcavit_amd.graph_topology import Topology, check


def chain(nodes):
    return [(a, b) for a, b in zip(nodes[:-1], nodes[1:])]


def test_a_plain_chain_and_an_empty_graph_pass():
    assert check(Topology(0, [])) == []
    assert check(Topology(1, [])) == []
    assert check(Topology(6, chain(list(range(6))))) == []


def test_one_fork_one_chain_one_join_passes_also_three_in_sequence():
    # main 0-1-2-3-4-5, side 6-7-8 forked behind 1, joined in front of 4
    e = chain([0, 1, 2, 3, 4, 5]) + [(1, 6)] + chain([6, 7, 8]) + [(8, 4)]
    assert check(Topology(9, e)) == []
    # three such diamonds one after the other, the second join node is also the third fork
    e = chain(list(range(10))) + [(0, 10), (10, 2)] + [(3, 11), (11, 12), (12, 5)] + [(5, 13), (13, 8)]
    assert check(Topology(14, e)) == []
    # duplicate edges are one dependency
    assert check(Topology(9, chain([0, 1, 2, 3, 4, 5]) + [(1, 6), (1, 6)] + chain([6, 7, 8]) + [(8, 4)])) == []


def test_redundant_stream_order_edges_are_ignored():
    """What the runtime really records for two forks in sequence on ONE side stream (tools/graph_shapes.py 'three_diamonds',
    profiles/r05_graph_shapes.txt): the side stream's own order adds an edge from the last launch of the first side chain to
    the first launch of the second, which already waits for the main chain behind the first join."""
    main = list(range(10))
    e = chain(main) + [(1, 10), (10, 11), (11, 3)] + [(5, 12), (12, 13), (13, 7)] + [(11, 12)]
    assert check(Topology(14, e)) == []
    # but a second REAL dependency of the side chain (nothing implies it) stays a violation
    assert check(Topology(14, chain(main) + [(1, 10), (10, 11), (11, 12), (12, 13), (13, 7)] + [(4, 12)]))


def test_a_side_stream_from_the_very_top_and_an_unjoined_tail_are_a_virtual_fork_and_join():
    # side chain 5-6 has no predecessor (forked before the main stream issued anything), joined into main node 3
    assert check(Topology(7, chain([0, 1, 2, 3, 4]) + [(5, 6), (6, 3)])) == []
    # a side chain that is never joined ends the graph beside the main chain
    assert check(Topology(6, chain([0, 1, 2, 3]) + [(1, 4), (4, 5)])) == []


def test_shape_2_a_side_chain_with_a_second_edge_from_the_main_chain_fails():
    # the object branch forked at the top (root 6), the skip convolutions forked behind main node 2 onto the SAME side stream:
    # side node 8 depends on side node 7 (stream order) and on main node 2
    e = chain([0, 1, 2, 3, 4, 5]) + chain([6, 7, 8, 9]) + [(2, 8), (9, 4)]
    v = check(Topology(10, e))
    assert v and ("second" in v[0] or "nested" in v[0] or "do not meet" in v[0]), v


def test_shape_1_four_forks_and_joins_between_two_streams_fails():
    main = list(range(12))
    side = [12, 13, 14, 15]
    e = chain(main) + chain(side) + [(1, 12), (3, 13), (5, 14)] + [(13, 6), (14, 8), (15, 10)]
    assert check(Topology(16, e))


def test_three_parallel_branches_and_a_nested_fork_fail():
    # three branches out of node 1
    e = chain([0, 1, 2, 3, 4]) + [(1, 5), (5, 3), (1, 6), (6, 3)]
    v = check(Topology(7, e))
    assert v and "more than two" in v[0], v
    # a fork inside an open fork: side node 5 forks again
    e = chain([0, 1, 2, 3, 4]) + [(1, 5), (5, 6), (6, 3), (5, 7), (7, 8), (8, 3)]
    v = check(Topology(9, e))
    assert v and "nested" in v[0], v


def test_violations_name_the_node():
    t = Topology(7, chain([0, 1, 2, 3, 4]) + [(1, 5), (5, 3), (1, 6), (6, 3)], types=["kernel"] * 7,
                 names=["k%d" % i for i in range(7)])
    assert "#1 kernel k1" in check(t)[0]
    s = t.summary()
    assert s["nodes"] == 7 and s["forks"] == 1 and s["joins"] == 1 and s["max_out"] == 3
