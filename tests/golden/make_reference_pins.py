"""Extract the hard-coded expected arrays (DATA only) from the reference's own
golden test, tests/test_reference_results.py:26-63,93-130, into a JSON fixture.

Run in the build container only (needs /root/reference); the fixture
tests/golden/reference_results.json is what travels to the GPU box.
"""
import ast
import json
import pathlib

SRC = pathlib.Path("/root/reference/tests/test_reference_results.py")
OUT = pathlib.Path(__file__).with_name("reference_results.json")


def main():
    tree = ast.parse(SRC.read_text())
    out = {}
    for fn in tree.body:
        if not isinstance(fn, ast.FunctionDef):
            continue
        case = "full" if "full" in fn.name else "sparse"
        for node in fn.body:
            if (isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name)
                    and node.targets[0].id.startswith("expected_")):
                arr = ast.literal_eval(node.value.args[0])
                out.setdefault(case, {})[node.targets[0].id] = arr
    out["meta"] = {
        "source": "settylab/Mellon v1.7.1 tests/test_reference_results.py:26-63,93-130",
        "inputs": "jax.random.PRNGKey(42) -> split(3) -> normal (50,2),(50,3),(10,2); "
                  "FunctionEstimator(sigma=1.0, n_landmarks=0|15)",
        "atol": 1e-5,
    }
    OUT.write_text(json.dumps(out, indent=1))
    print("wrote", OUT, {k: list(v) for k, v in out.items() if k != "meta"})


if __name__ == "__main__":
    main()
