"""The shipped default is what is tested: the golden / oracle-anchored estimator tests once more in a process WITHOUT
MELLON_AMD_EXPERIMENTAL (the rest of the suite sets it so that tests can turn experiment knobs; the library reads the switch
once per process, csrc/mln_options.h)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SELECT = ("test_density_golden_fixtures or test_time_sensitive_golden_fixture or test_function_estimator_reference_golden or "
          "test_reference_golden_leverage_and_obs_variance or test_c3_subsample_golden or test_c2_scaled_expquad or "
          "test_c4_shaped_against_oracle or test_c5_shaped_against_oracle or test_reference_golden_function_estimator_on_gpu or "
          "test_c2_full_size_against_oracle")


def test_golden_tests_without_the_experiment_switch():
    env = {k: v for k, v in os.environ.items() if not k.startswith("MELLON_AMD_")}
    env["MELLON_AMD_TEST_SHIPPED_DEFAULTS"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_estimators.py"),
                          os.path.join(ROOT, "tests", "test_gpu_ops.py"), os.path.join(ROOT, "tests", "test_gpu_configs.py"),
                          "-m", "gpu", "-q", "-x", "-k", SELECT, "-p", "no:cacheprovider"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = run.stdout[-3000:]
    assert run.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail.split("\n")[-2], tail
