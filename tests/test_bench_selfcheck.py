"""``bench.roofline_self_check``: the line's own plausibility test for its roofline figures (review r04, row d).

Round 4's driver-run line said ``kernel_ms_per_batch`` 17.07 inside an 18.6-ms batch (``frac`` 0.255 where the rocprofv3 summary
said 0.488): the clock sampler's stream had been serialised in front of the traced launch.  The checks must flag exactly that
line and pass this round's."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                 # (defines functions only: main() runs under __main__)
    return mod


def test_round4_driver_line_is_flagged():
    b = _bench()
    # BENCH_r04: block sums 17.07 ms "in" an 18.6-ms batch; the 16-class launch 16.385 ms against 6.9 ms back to back at 2 397 MHz
    chk = b.roofline_self_check(17.07, 5.6, [16.385], [18.5, 18.6, 18.7, 18.6, 18.9], 6.9, 2397.0)
    assert chk["fits_in_batch"] is False and chk["class_launch_plausible"] is False and chk["ok"] is False
    # either symptom alone is enough
    assert b.roofline_self_check(17.07, 5.6, [], [18.6], None, None)["ok"] is False
    assert b.roofline_self_check(8.9, 5.6, [16.4], [18.6], 6.9, 2397.0)["ok"] is False


def test_round5_lines_pass():
    b = _bench()
    chk = b.roofline_self_check(8.977, 5.38, [8.297], [18.3, 18.4, 18.2, 18.3, 18.9], 7.45, 2010.0)
    assert chk["ok"] and chk["fits_in_batch"] and chk["class_launch_plausible"]
    assert chk["blocksum_plus_chain_ms"] == round(8.977 + 5.38, 3)
    # no clock sample (sampler discarded or several ranks): the in-situ clock of every run so far stands in
    assert b.roofline_self_check(8.977, 5.38, [8.297], [18.3], 7.45, None)["ok"]
    # nothing measured (--no-roofline-batch): nothing to object to
    assert b.roofline_self_check(0.0, 0.0, [], [], None, None)["ok"]
