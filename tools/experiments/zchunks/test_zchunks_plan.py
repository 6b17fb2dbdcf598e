"""The communication plan of the step pipelined along z (latticeurbanwind_amd/zchunks.py comm_plan; LUW_STEP_SCHEDULE=zchunks, an experiment behind its switch):
pure logic, no GPU.  What the schedule's correctness rests on, for every chunk count and every rotation of the start:
  * every chunk is sent once, after it was computed, and inserted once, after its own faces were sent and after all its z neighbours were computed
    (they still read what the previous step put there);
  * what the FIRST chunk of the next step waits for -- the inserts of its three neighbours -- is enqueued before the last chunk's faces go on the wire."""
import pytest

from latticeurbanwind_amd.zchunks import comm_plan, neighbours


@pytest.mark.parametrize("C", [4, 5, 6, 8])
def test_plan_properties(C):
    for start in range(C):
        order = [(start + j) % C for j in range(C)]
        plan = comm_plan(order, C)
        pos = {ev: i for i, ev in enumerate(plan)}
        assert len(pos) == len(plan) == 3 * C                                   # wait, send, insert: once per chunk
        for k in range(C):
            assert pos[("wait", k)] < pos[("send", k)] < pos[("insert", k)]
            for m in neighbours(k, C):
                assert pos[("wait", m)] < pos[("insert", k)], (C, start, k, m)
        # waits follow the compute order (the communication stream is in order: a wait for chunk k covers the chunks computed before it)
        assert [k for what, k in plan if what == "wait"] == order
        # the next step starts one chunk further: its first chunk needs the inserts of that chunk's neighbours -- none of them behind the last send
        first_next = (start + 1) % C
        last_send = pos[("send", order[-1])]
        assert all(pos[("insert", m)] < last_send for m in neighbours(first_next, C)), (C, start)


def test_neighbours_wrap():
    assert neighbours(0, 4) == {3, 0, 1} and neighbours(3, 4) == {2, 3, 0}
