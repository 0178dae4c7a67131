"""CPU oracle for the Matsuno C-grid hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A unit-free float64 NumPy restatement of the reference's algorithm (every
function cites the /root/reference file:line it follows) with the reference's
own operation order, SI inputs.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package; nothing
under ``gcmiipy_amd/`` does, and the product path raises when the HIP library
is missing instead of falling back to this code.

Pinning (how we know this oracle equals the reference):
  * the reference's own known-answer tests for the path (test_matsumo.py:9-29,
    test_2d.py:176-181, flux_limiter.py:46-48, temperature.py:31-41, the TV
    bounds of test_2d.py:47-80,140-173), restated in tests/test_oracle_kats.py;
  * golden vectors in tests/golden/*.npz produced by tests/golden/make_golden.py,
    which imports the UNMODIFIED reference modules from /root/reference inside
    the build container and runs them on seeded inputs.  The image has no
    `pint`, so that script puts a unit-bookkeeping stand-in
    (tests/golden/_pint_standin/pint.py) on sys.path; the arithmetic executed is
    the reference's own NumPy expressions.  The oracle matches those vectors
    BIT FOR BIT (tests/test_oracle_golden.py).
  * the composed van-Leer-limited 2-D tracer step (oracle/tracer.py
    `limited_*`) has no counterpart in the reference: parity
    UNPINNED for that composition; its pieces (phi, r, donor flux, upwind and
    centred steps) are pinned as above.

Unit literals the reference carries through pint and that are folded here:
dx arrives in metres, mu_air = 18.5 uPa s -> 18.5*1e-6, ptop = 0 hPa -> 0.0.
"""
