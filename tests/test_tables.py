"""Host logic of the product library (no GPU): the C++ pattern-table builder behind gmk_tables_* must
reproduce the reference automaton -- checked against the oracle's restatement of it -- and the
flattened DFA the kernel walks must emit the reference's match stream."""
import numpy as np
import pytest

from gomokuai_amd import lib as G


def test_dat_and_patterns_equal_oracle(oracle):
    O = oracle
    ac = O.default_ac()
    info = G.tables_info()
    assert (info.n_patterns, info.dat_size) == (294, 1024)
    assert list(info.invariants) == list(ac.invariants)
    base, check, fail = G.tables_copy_dat()
    assert (base == np.array(ac.base[:1024])).all()
    assert (check == np.array(ac.check[:1024])).all()
    assert (fail == np.array(ac.fail[:1024])).all()
    for i in range(294):
        p = ac.patterns[i]
        assert G.tables_pattern(i) == (p.str.decode(), p.favour, p.type, p.score)


def test_table_sizes_fit_lds():
    info = G.tables_info()
    assert info.n_states <= 1024 and info.emit_words < 1024 * 2
    lds = info.trans_words * 4 + info.emit_words * 2 + info.n_patterns * 8
    assert lds < 16 * 1024          # staged once per workgroup
    assert info.max_emissions <= 3


def test_pattern_info_words(oracle):
    ac = oracle.default_ac()
    _, _, pinfo = G.tables_copy()
    for i in range(294):
        p = ac.patterns[i]
        w0, w1 = int(pinfo[2 * i]), int(pinfo[2 * i + 1])
        assert w0 & 15 == p.type and ((w0 >> 4) & 1) == (p.favour == 1) and ((w0 >> 5) & 7) == p.len
        s = p.str.decode()
        for j in range(p.len):
            kind = (w0 >> (8 + 2 * j)) & 3
            piece = s[p.len - 1 - j]
            assert kind == (0 if p.type == 8 else {"_": 1, "^": 2}.get(piece, 0))
        assert w1 & 0xffff == p.score and w1 >> 16 == int(1.2 * p.score) == (6 * p.score) // 5


def _streams(O, ac, line):
    a = [(p.type, p.str, off) for p, off in O.match(ac, line)]
    g = [(ac.patterns[i].type, ac.patterns[i].str, off) for i, off in G.tables_scan(line)]
    return a, g


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_flattened_dfa_equals_reference_generator(oracle, seed):
    """Same (pattern, end offset) stream, in the same order, on random padded lines.  `Five` is compared as
    a set: the reference reports it once per run end (Pattern.cpp:40-45), the DFA at every stone past the
    fifth, and only its presence is ever used (Pattern.cpp:140-145)."""
    O = oracle
    ac = O.default_ac()
    rng = np.random.RandomState(seed)
    for _ in range(6000):
        n = rng.randint(1, 28)
        cells = rng.choice([1, 2, 4], size=n, p=[0.3, 0.3, 0.4]).astype(np.uint8)
        line = np.concatenate([np.full(rng.randint(0, 7), 3, np.uint8), cells, np.full(rng.randint(0, 7), 3, np.uint8)])
        a, g = _streams(O, ac, line)
        assert [x for x in a if x[0] != 8] == [x for x in g if x[0] != 8]
        assert {x[1] for x in a if x[0] == 8} == {x[1] for x in g if x[0] == 8}


def test_reference_match_stream_kat_through_product_tables(oracle):
    """patternsearch_unittest.cpp:204-223 uses a 3-pattern table; on the production table the same target
    must still contain those five matches (they are production patterns too)."""
    O = oracle
    ac = O.default_ac()
    line = O.encode("??-xxx-ooo-xxx-o-xxx--xxx-?")
    got = {(tuple(O.encode(ac.patterns[i].str.decode())), off) for i, off in G.tables_scan(line)}
    ref = {(tuple(O.encode(p.str.decode())), off) for p, off in O.match(ac, line)}
    assert got == ref


def test_unreachable_patterns_quirk(oracle):
    """Three patterns lose their trie path to the std::sort tie order (see pattern_tables.cpp): neither the
    oracle nor the product may ever report them."""
    O = oracle
    ac = O.default_ac()
    lost = {"~x__xx~", "~o__oo~", "~_x~xx~"}
    for s in lost:
        line = np.concatenate([[3], O.encode(s), [3, 3]]).astype(np.uint8)
        assert s not in {ac.patterns[i].str.decode() for i, _ in G.tables_scan(line)}
        assert s not in {p.str.decode() for p, _ in O.match(ac, line)}
