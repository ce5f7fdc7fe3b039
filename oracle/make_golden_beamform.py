#!/usr/bin/env python3
"""Generate beamformer golden fixtures under tests/golden/ with the reference's
own CPU beamformer functions.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).

The reference's software beamformer lives in
  pipeline/verification/tests/beamformer/beamformer_test.py   (SoftwareBf.get_input_gulp :61-74, .cpu_beamform :76-84)
  pipeline/verification/tests/beamformer/beamformer_sum_test.py (SoftwareBfSum.cpu_sum_power :64-77)
Both files `import bifrost` at module top, which is not installed here (ordinary
ModuleNotFoundError), so the modules cannot be imported whole.  The three
functions themselves are pure numpy: this script parses the files with `ast`,
compiles ONLY those function definitions, and calls them on seeded inputs.  No
reference source text is written into the repo -- only inputs and the outputs
the reference functions computed (tests/golden/beamform_*.npz).
"""
import ast
import io
import os
import types

import numpy as np

REFDIR = "/root/reference/pipeline/verification/tests/beamformer"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def extract(path, names):
    """Compile the named FunctionDefs found anywhere in `path` and return them."""
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"np": np}
            exec(compile(mod, path, "exec"), ns)
            found[node.name] = ns[node.name]
    missing = set(names) - set(found)
    assert not missing, missing
    return found


def main():
    f1 = extract(os.path.join(REFDIR, "beamformer_test.py"), ["get_input_gulp", "cpu_beamform"])
    f2 = extract(os.path.join(REFDIR, "beamformer_sum_test.py"), ["cpu_sum_power"])

    for tag, (ntime, nchan, nstand, nbeam, ntime_sum) in {
            "small": (48, 4, 20, 4, 12),
            "tile": (96, 3, 32, 32, 24)}.items():
        ninput = nstand * 2
        rng = np.random.default_rng(0xaabbccdd)  # beamformer_test.py:121
        vin = rng.integers(0, 256, size=(ntime, nchan, ninput), dtype=np.uint8)
        # weights: random complex64, including the magnitudes the block produces (amps 10..17)
        w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) +
             1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
        raw = vin.tobytes()
        obj = types.SimpleNamespace(ntime_gulp=ntime, nchan=nchan, ninput=ninput, nbeam=nbeam,
                                    coeffs=w, testfile=io.BytesIO(raw), testfile_nbytes=len(raw))
        decoded = f1["get_input_gulp"](obj, 0)          # [time, chan, input] complex
        beams = f1["cpu_beamform"](obj, decoded)        # complex64 [chan, beam, time]
        power = f2["cpu_sum_power"](obj, beams, ntime_sum)  # f32 [beam/2, time/ntime_sum, chan, 4]
        np.savez_compressed(os.path.join(GOLD, "beamform_%s.npz" % tag),
                            vin=vin, weights=w, decoded=np.asarray(decoded, dtype=np.complex64),
                            beams=beams, power=power, ntime_sum=np.int32(ntime_sum))
        print(tag, beams.shape, power.shape, float(np.abs(beams).max()))


if __name__ == "__main__":
    main()
