"""The transform passes' LDS layout against the bank model (tools/lds_model.py): a guard for round 5's padding fix.

`k_ntt29_pass` keeps a tile of nine-word element records in LDS and reaches them with ds_read2_b32 / ds_write_b32, which
the MI355X services a 32-lane half at a time over 32 banks.  The padding that suited a 64-bank picture (a word per 16
elements, rounds 1-5) put two lanes of every half on one bank in every phase - SQ_LDS_BANK_CONFLICT read half of
SQ_LDS_IDX_ACTIVE (profiles/r05_am_lds_bank_conflicts.md).  The model reproduces both figures; this test reads the
padding out of the shipped source and holds it to the modelled optimum."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("lds_model", os.path.join(ROOT, "tools", "lds_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_transform_tile_padding_is_the_modelled_optimum(capsys):
    src = open(os.path.join(ROOT, "crescent-credentials_amd", "csrc", "wmap29.hip")).read()
    m = re.search(r"lds_off\(uint32_t e\) \{ return e \* 9u \+ \(e >> (\d+)\); \}", src)
    assert m, "lds_off() changed shape: re-derive its padding with tools/lds_model.py"
    shift = int(m.group(1))
    mod = _model()
    capsys.readouterr()
    shapes = [mod.rows_for((0, 2, 4, 6, 8), 10), mod.rows_for((1, 3, 5, 7, 9)), mod.rows_for((0, 2, 4, 6, 8), None, 1024, 256),
              mod.rows_for((1, 3, 5, 7), 9, 1024, 256)]
    for rows in shapes:
        now = mod.score(rows, lambda E: E >> shift)
        old = mod.score(rows, lambda E: E >> 4)
        assert sum(now.values()) / len(now) <= 0.34, now           # two-way conflicts in at most two of the stage pairs
        assert now["linear"] == 0.0                                # the load and store phases are conflict-free
        assert sum(old.values()) / len(old) == 1.0                 # what the counters showed for rounds 1-5
    # the records still fit the array the kernel declares: (1 << TSL) * 9 + (1 << TSL) / 16 words
    for tsl in (10, 11):
        n = 1 << tsl
        assert (n - 1) * 9 + ((n - 1) >> shift) + 9 <= n * 9 + n // 16
