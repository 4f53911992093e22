"""Self-play records and the multi-rank gather (CPU part).  The gather is the only exchange step of the path
(SURVEY.md 8e); it is exercised here with world_size 2 on gloo, the GPU box uses the same code over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gomokuai_amd import core, selfplay


def _fake_records(n, first, seed):
    rng = np.random.RandomState(seed)
    moves = np.zeros((n, 225), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    winner = np.zeros(n, dtype=np.int8)
    visits = np.zeros((n, 225, 225), dtype=np.int16)
    for g in range(n):
        b = core.Board()
        t = 0
        while not b.status["is_end"]:
            mv = b.random_move()
            empties = [i for i in range(225) if b.check_move(core.Position(i))]
            visits[g, t, empties] = rng.randint(0, 50, size=len(empties))
            visits[g, t, mv.id] = 60
            moves[g, t] = mv.id
            b.apply_move(mv)
            t += 1
        lens[g] = t
        winner[g] = int(float(b.status["winner"]))
    return selfplay.GameRecords(torch.from_numpy(moves), torch.from_numpy(lens), torch.from_numpy(winner), torch.from_numpy(visits), first)


def test_samples_match_dual_play_tuples():
    """GameRecords.samples == what agents/utils.py:29-63 builds move by move from Board.encoded_states()."""
    rec = _fake_records(2, 0, 1)
    for g in range(2):
        samples = rec.samples(g)
        b = core.Board()
        L = int(rec.lens[g])
        assert len(samples) == L
        for t in range(L):
            states, score, pi = samples[t]
            assert (states == b.encoded_states()).all()
            cur = b.status["cur_player"]
            b.apply_move(core.Position(int(rec.moves[g, t])))
            assert pi.shape == (225,) and abs(float(pi.sum()) - 1.0) < 1e-3
            assert float(score) == core.Player.calc_score(cur, core.Player.black if rec.winner[g] == 1 else core.Player.white if rec.winner[g] == -1 else core.Player.none)
        assert b.status["is_end"]


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tuples.npz")


def test_host_tuples_match_the_reference_loop():
    """tests/golden/reference_tuples.npz holds what the REFERENCE's dual_play(verbose=True) (agents/utils.py:29-63) produced on
    this CorePyExt for six games of its random agent (made by tests/golden/make_reference_tuples.py in the build container).
    GameRecords.samples rebuilds the tuples from the move lists alone: encoded states and values must be identical."""
    ref = np.load(GOLDEN)
    n = int(ref["lens"].shape[0])
    rec = selfplay.GameRecords(torch.from_numpy(ref["moves"]), torch.from_numpy(ref["lens"]), torch.from_numpy(ref["winner"]),
                               torch.zeros((n, 225, 225), dtype=torch.int16), 0)
    k = 0
    for g in range(n):
        for states, value, pi in rec.samples(g):
            assert int(ref["game"][k]) == g
            assert (states == ref["states"][k]).all(), "game %d tuple %d" % (g, k)
            assert float(value) == float(ref["values"][k])
            assert pi.shape == ref["probs"][k].shape
            k += 1
    assert k == int(ref["values"].shape[0]) == int(ref["lens"].sum())
    assert np.allclose(ref["probs"], 1.0 / 225)                # the reference's random agent reports the uniform distribution


def test_shard_blocks_cover_everything():
    for total, world in ((32768, 8), (4096, 3), (10, 4)):
        blocks = [selfplay.shard(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
        for (lo, n), (lo2, _) in zip(blocks, blocks[1:]):
            assert lo + n == lo2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gather_worker(rank, world, port, out_dir, total, dst):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, n = selfplay.shard(total, rank, world)
    rec = _fake_records(n, lo, 100 + rank)
    rec.overflow = rank == world - 1                       # one rank saw an arena overflow: rank dst must hear of it
    got = selfplay.gather_records(rec, dst=dst)
    torch.save({"moves": rec.moves, "lens": rec.lens, "winner": rec.winner, "visits": rec.visits}, os.path.join(out_dir, "part%d.pt" % rank))
    if rank == dst:
        assert got is not None and len(got) == total and got.first_game_id == 0 and got.overflow
        torch.save({"moves": got.moves, "lens": got.lens, "winner": got.winner, "visits": got.visits}, os.path.join(out_dir, "all.pt"))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,dst", [(2, 7, 0), (3, 7, 1), (3, 2, 0)])      # even-ish shards; uneven shards to a non-zero root; a rank without a game
def test_gather_gloo(tmp_path, world, total, dst):
    """The exchange step with world size 2 and 3 on gloo: ONE code path for every backend (grouped point-to-point transfers of
    the compact wire form), so this is what RCCL runs on the GPU box."""
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path), total, dst), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, "part%d.pt" % r)) for r in range(world)]
    allrec = torch.load(os.path.join(tmp_path, "all.pt"))
    for key in ("moves", "lens", "winner", "visits"):
        assert torch.equal(allrec[key], torch.cat([p[key] for p in parts], dim=0)), key


def _failing_gather_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    lo, n = selfplay.shard(6, rank, world)
    rec = _fake_records(n, lo, 7 + rank)
    if rank == 1:
        rec.moves = None                                   # packing this rank's records raises before the first collective
    try:
        selfplay.gather_records(rec, dst=0)
        outcome = "returned"
    except selfplay.GatherError as exc:
        outcome = "GatherError: %s" % exc
    open(os.path.join(out_dir, "outcome%d.txt" % rank), "w").write(outcome)
    dist.barrier()                                         # every rank is still in step with the others
    dist.destroy_process_group()


def test_gather_fails_on_every_rank_when_one_cannot_pack(tmp_path):
    """A rank that fails before all_gather must not leave the others waiting in it: the ranks agree first, all raise GatherError."""
    import time
    t0 = time.monotonic()
    mp.spawn(_failing_gather_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert time.monotonic() - t0 < 45.0
    outcomes = [open(os.path.join(tmp_path, "outcome%d.txt" % r)).read() for r in range(3)]
    assert all(o.startswith("GatherError") for o in outcomes), outcomes
    assert "this rank" in outcomes[1] and "another rank" in outcomes[0]


def test_wire_form_is_compact_and_round_trips():
    """pack_records carries only what was played (moves u8[L], visits u16[L][225], winner, length per game)."""
    rec = _fake_records(3, 9, 5)
    buf = selfplay.pack_records(rec)
    total = int(rec.lens.sum())
    assert buf.dtype == torch.uint8 and buf.numel() == 3 * 5 + total + total * 225 * 2
    assert buf.numel() < 0.6 * (rec.moves.numel() + rec.visits.numel() * 2)
    back = selfplay.unpack_records(buf, 3, True, first_game_id=9)
    for key in ("moves", "lens", "winner", "visits"):
        assert torch.equal(getattr(back, key), getattr(rec, key)), key
    no_visits = selfplay.GameRecords(rec.moves, rec.lens, rec.winner, None, 9)
    back = selfplay.unpack_records(selfplay.pack_records(no_visits), 3, False)
    assert back.visits is None and torch.equal(back.moves, rec.moves)


def test_gather_single_process_is_identity():
    rec = _fake_records(2, 5, 3)
    assert selfplay.gather_records(rec) is rec


def test_host_games_match_the_board():
    """selfplay._HostGames (all games of a batch at once, numpy) against CorePyExt.Board move by move: end detection and winner
    (Game.cpp:37-49, 88-136), including overlines and a full board without a five (the reference's tie order)."""
    from gomokuai_amd import core
    from gomokuai_amd.selfplay import _HostGames
    rng = np.random.RandomState(7)
    n = 120
    hg = _HostGames(n)
    boards = [core.Board() for _ in range(n)]
    for ply in range(225):
        played = np.full(n, -1)
        for g in range(n):
            if boards[g].status["is_end"]:
                continue
            free = [c for c in range(225) if boards[g].check_move(core.Position(c))]
            played[g] = free[rng.randint(len(free))] if (g % 3 and rng.rand() < 0.6) else free[0]
        moved = hg.apply(played)
        assert sorted(moved) == [g for g in range(n) if played[g] >= 0]
        for g in range(n):
            if played[g] >= 0:
                boards[g].apply_move(core.Position(int(played[g])))
        assert all(bool(hg.over[g]) == bool(boards[g].status["is_end"]) for g in range(n)), ply
        if hg.over.all():
            break
    assert all(int(hg.winner[g]) == int(boards[g].status["winner"]) for g in range(n))
    assert (hg.winner == 1).any() and (hg.winner == -1).any()
    # the tie: a full board without five in a row (the pattern of core/test/integration/board_integrationtest.cpp)
    tie = _HostGames(1)
    b = core.Board()
    order = [y * 15 + x for y in range(15) for x in range(15)]
    cells = sorted(order, key=lambda c: (((c % 15) // 2 + (c // 15)) % 2, c))      # colour classes that never line up five
    blacks, whites = [c for c in cells if ((c % 15) // 2 + c // 15) % 2 == 0], [c for c in cells if ((c % 15) // 2 + c // 15) % 2 == 1]
    seq = []
    while blacks or whites:
        if blacks: seq.append(blacks.pop())
        if whites: seq.append(whites.pop())
    ok = True
    for c in seq:
        if b.status["is_end"]:
            ok = False
            break
        tie.apply(np.array([c]))
        b.apply_move(core.Position(c))
        assert bool(tie.over[0]) == bool(b.status["is_end"])
    assert int(tie.winner[0]) == int(b.status["winner"])


def test_plan_games_spreads_games_over_handles_and_slots():
    """selfplay.plan_games: the bookkeeping behind play_games(slots=, handles=) -- contiguous blocks that cover every game once, the
    slots shared out between the handles, the lock-step loop when neither is asked for."""
    assert selfplay.plan_games(4096) is None                                    # one handle, all games at once
    assert selfplay.plan_games(23, slots=64) is None                            # more slots than games
    assert selfplay.plan_games(32768) == [(0, 32768, 8192)]                     # one handle: its persistent launch fills the chip
    assert selfplay.plan_games(8192) is None                                    # every game in flight from the start
    assert selfplay.plan_games(8192, handles=2) == [(0, 4096, 4096), (4096, 8192, 4096)]   # two handles on request
    assert selfplay.plan_games(23, slots=5) == [(0, 23, 5)]
    for n, slots, handles in [(23, 6, 3), (23, None, 2), (7, 100, 7), (5, 1, 9), (40000, "auto", "auto"), (16385, "auto", 1)]:
        plan = selfplay.plan_games(n, slots, handles)
        assert plan[0][0] == 0 and plan[-1][1] == n and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        assert all(1 <= s <= hi - lo for lo, hi, s in plan)
        if isinstance(slots, int):
            assert sum(s for _, _, s in plan) <= max(len(plan), min(slots, n)) + len(plan) - 1
    assert selfplay.plan_games(20000, opening_plies=12) is None                 # long openings: the lock-step loop
    with pytest.raises(ValueError):
        selfplay.plan_games(20000, slots=100, opening_plies=12)
    with pytest.raises(ValueError):
        selfplay.plan_games(0)


def test_network_slot_loops_refuse_what_they_would_ignore():
    """play_network_games: the slot loops play whole games (and the device-resident one takes openings of at most 8 plies): a move cap or a longer
    opening is refused up front, not silently ignored (no GPU needed: the check comes first)."""
    import pytest
    from gomokuai_amd import selfplay
    with pytest.raises(ValueError):
        selfplay.play_network_games(8, None, 4, slots=2, max_moves=10)
    with pytest.raises(ValueError):
        selfplay.play_network_games(8, None, 4, slots=2, opening_plies=12)
