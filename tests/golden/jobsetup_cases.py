"""The cases of tests/golden/jobsetup.npz: numbers only.  Shared by make_fixtures.py (which runs the reference on them) and
the tests (which rebuild the same inputs from the seeds)."""

# (tag, content (h, w, seed), style (h, w, seed), numpy seed, Config kwargs): three init methods; landscape, portrait
# and square content; positive / negative / zero granularities; 1-3 levels.  tests/test_oracle_jobsetup.py rebuilds the
# inputs from these numbers.
JOBSETUP_CASES = (
    ("landscape_default", (200, 300, 1), (230, 170, 2), 0, dict(levels_num=2)),
    ("portrait_random", (300, 200, 3), (170, 230, 4), 5,
     dict(levels_num=2, init_method="random", noise_levels=(5, -3, 0, -16),
          noise_levels_central_amplitude=(0.3, 0.25, 0.1, 0.2), noise_levels_peripheral_amplitude=(0.1, 0.3, 0.05, 0.4),
          noise_levels_dispersion=(0.25, 0.5, 0.3, 0.7), noise_factor=0.8)),
    ("square_style", (180, 180, 5), (100, 140, 6), 7, dict(levels_num=1, init_method="style")),
    ("wide_three_levels", (120, 330, 7), (150, 90, 8), 11,
     dict(levels_num=3, noise_levels=(4, -2, 0), noise_levels_central_amplitude=(0.4, 0.2, 0.1),
          noise_levels_peripheral_amplitude=(0.1, 0.5, 0.0), noise_levels_dispersion=(0.3, 0.6, 0.2), noise_factor=0.5)),
    ("one_level_default", (97, 131, 9), (64, 64, 10), 13, dict(levels_num=1)),
)
