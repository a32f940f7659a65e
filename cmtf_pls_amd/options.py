"""`EngineOptions`: which exact form of each step the engine takes where the shape allows it (one object per engine)."""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional


@dataclass(frozen=True)
class EngineOptions:
    """Which exact form of each step the engine takes WHERE THE SHAPE ALLOWS IT.  Every default is the fastest form; each
    switch selects the slower equivalent form the tests compare it with.  One object per engine (`NipalsEngine(backend,
    comm, options)`, `tPLS(..., options=EngineOptions(...))`); what actually ran is written to `FitState.report`
    (`tPLS.fit_report_`), so a path the shape declined is visible instead of silent."""
    # a single small float64 block without missing values: the whole fit in ONE launch (cmtfpls_fit_small_f64); a regular
    # iteration is ~20 launches of pure latency whatever the size, a one-workgroup iteration costs time in proportion to I * P
    small_fit: bool = True
    small_fit_elements: int = 1 << 15    # measured (profiles/r03q_small_fit.txt): 2.2x faster at 16000 elements, slower from 65536 on
    # algorithm="xcov" on blocks without missing values: never deflate X in place (two reads per component instead of a
    # read and a read + write, FitRun._finish_xcov_nowrite); False keeps the deflating form
    xcov_nowrite: bool = True
    # ... and, when X is never written anyway, do not centre it either: the fit runs on the caller's UNCENTRED tensor -- no
    # centring pass, no private copy -- with two rank-one corrections; False keeps the centred copy
    xcov_raw: bool = True
    # the uncentred form works by cancellation: its error grows with max|column mean| / rms spread of the centred data.
    # Beyond this ratio the fit falls back to the centred private copy (report: raw = False, raw_declined = ratio)
    xcov_raw_max_offset: float = 1e4
    # the largest block: score and the contraction with the (block-averaged) score from ONE read of it, the second read per
    # component replaced by a P x a matrix-vector product; False keeps the two reads
    xcov_one_read: bool = True
    # blocks WITH missing values, 2 M <= 64: S = X0^T Y and S2 = X0^T (Y * rowscale) from one matrix-core pass with the
    # I x 2M right-hand side [Y, Y * rowscale]; False builds them one after the other
    xcov_pair_build: bool = True
    # a fit on the uncentred tensor: |X - X_mean|^2 from the read that builds S for the first component instead of a read of
    # its own (backend.xcov_ssq); False keeps the separate pass
    xcov_ssq_with_s: bool = True
    # ... and the statistics pass too: column sums and sums of squares from the read that builds S for the first component
    # (backend.xcov_stats): a fit on the uncentred tensor reads X once before its first component; False keeps colstats first
    xcov_stats_with_s: bool = True
    # one block WITH missing values: the deflation happens inside the rebuild of S for the next component (one read + write
    # of X instead of a read + write and a read, FitRun._finish_xcov_masked_fused); False keeps the two passes
    xcov_deflate_build: bool = True
    # the inner loop on S: iteration it + 1 is ENQUEUED before the host has seen iteration it's convergence norm, into a
    # second set of buffers (FitRun._inner_loop_xcov_pipelined); False waits after every iteration
    xcov_pipeline: bool = True
    # sharded direct loop under graph replay: capture the two per-iteration all-reduces INSIDE the iteration's HIP graph (one
    # replay per iteration instead of three segments and two eager collectives); falls back to the segment-wise form when
    # the capture fails (report: collectives_in_graph)
    capture_collectives: bool = False
    # transform / predict of samples with missing values: rows WITHOUT a missing value keep the one-pass MTTKRP result and
    # only the affected rows take the masked sequential form; False runs the sequential form on every row of such a batch
    project_split_rows: bool = True

    def but(self, **changes) -> "EngineOptions":
        return replace(self, **changes)


_DEFAULT_OPTIONS = EngineOptions()


def default_options() -> EngineOptions:
    """The options of an engine constructed without any (the product default: `EngineOptions()`)."""
    return _DEFAULT_OPTIONS


def set_default_options(options: Optional[EngineOptions]) -> EngineOptions:
    """Replace the process-wide default (None restores `EngineOptions()`); returns the previous one.  The test harness uses it
    to keep the small float64 fits of the kernel suites on the multi-launch engine (tests/conftest.py)."""
    global _DEFAULT_OPTIONS
    old, _DEFAULT_OPTIONS = _DEFAULT_OPTIONS, (options if options is not None else EngineOptions())
    return old
