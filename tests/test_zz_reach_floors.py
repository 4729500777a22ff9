"""Runs last (file name): a floor under the number of sweep cases that REACH each kernel form.

The sweeps skip the cases whose plan does not reach the form under test -- legitimate, and invisible: 104 skips at the end of
round 5.  A rule edit that sends plans elsewhere would turn passes into skips without a word.  tests/conftest.py counts passed /
skipped cases per test function; where a function ran completely in this session, fewer passes than its floor is a failure.
(The floors are the counts of round 6, profiles/round6/test_tally.json.)"""
import pytest

import conftest

FLOORS = {
    # test function: (cases in all, floor under the cases that reach the form = pass)
    "tests/test_benchmarked_instances.py::test_seeded_sweeps_on_full_height_tiles": (144, 52),   # plans with a periodic table (quad forms on full-height tiles)
    "tests/test_colpair.py::test_border_columns_on_column_pairs_match_the_oracle_and_the_other_column_kernels": (30, 26),   # ewa_colpair_kernel
    "tests/test_edge_columns.py::test_edge_columns_match_the_oracle_and_the_border_kernels": (36, 34),   # border columns inside the interior kernel's edge tiles
    "tests/test_rowpair_rows.py::test_border_rows_on_the_pair_kernel_match_the_oracle_and_the_row_strips": (20, 18),   # border rows on ewa_periodic_rowpair_kernel
    "tests/test_strip_kernel.py::test_strip_kernel_matches_oracle_and_the_other_border_forms": (26, 22),   # ewa_strip_kernel over rows and columns
}


@pytest.mark.gpu
@pytest.mark.parametrize("fn", sorted(FLOORS), ids=lambda f: f.split("::")[1][:48])
def test_enough_sweep_cases_reach_the_form(fn):
    total, floor = FLOORS[fn]
    rec = conftest.OUTCOMES.get(fn)
    if rec is None or rec.get("passed", 0) + rec.get("skipped", 0) + rec.get("failed", 0) < total:
        pytest.skip("the sweep did not run (completely) in this session")
    assert rec["passed"] >= floor, f"only {rec['passed']} of {total} cases reached the form (floor {floor}): {rec}"
