"""Mutated CU batches through the whole backend ON THE GPU (the counterpart of fuzz_builder.py): what xgpu_batch_create accepts must run through every kernel without
a fault or a hang, and the device must decode the unmutated picture to its golden afterwards.   usage: fuzz_gpu_batches.py <seed> [iterations]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
import golden_io


def main():
    rng = np.random.default_rng(int(sys.argv[1]))
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    names = list(golden_io.PICTURE_CASES)
    ran = refused = checked = 0
    for it in range(iters):
        name = names[rng.integers(len(names))]
        case, exp = golden_io.load_picture_case(name)
        b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in case["batch"].items()}
        fields = [k for k, v in b.items() if isinstance(v, np.ndarray) and v.size and k != "coef"]
        for _ in range(int(rng.integers(1, 4))):
            flat = b[fields[rng.integers(len(fields))]].reshape(-1)
            idx, lim, kind = int(rng.integers(flat.size)), np.iinfo(flat.dtype), rng.integers(4)
            if kind == 0:
                flat[idx] = lim.max
            elif kind == 1:
                flat[idx] = lim.min
            elif kind == 2:
                flat[idx] = rng.integers(lim.min, lim.max, endpoint=True)
            else:
                flat[idx] = np.clip(int(flat[idx]) + int(rng.integers(-8, 9)), lim.min, lim.max)
        print(it, name, flush=True)
        try:
            cases.run_gpu(dict(case, batch=b), ahead=bool(rng.integers(2)))
            ran += 1
        except Exception as e:      # noqa: BLE001 - a refusal (invalid argument) is the expected answer to most mutations
            if "-101" not in str(e) and "invalid" not in str(e).lower():
                print(f"iteration {it} ({name}): {e!r}")
                return 1
            refused += 1
        if it % 10 == 9:             # the device still decodes the untouched picture
            out = cases.run_gpu(case)
            if not all(np.array_equal(out[c], exp["out"][c]) for c in range(3)):
                print(f"iteration {it}: {name} no longer decodes to its golden")
                return 1
            checked += 1
    print(f"done: {ran} decoded, {refused} refused, {checked} golden checks")
    return 0


if __name__ == "__main__":
    sys.exit(main())
