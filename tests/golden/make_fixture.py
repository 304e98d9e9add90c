"""Dumps the three known-answer literals recorded in SURVEY.md section 8c into survey_8c_kat.json (NOT a generator: no
reference run; the generated vectors are make_reference_vectors.py -> reference_flux_vectors.npz)."""
import json
import os

KAT = {
    "source": "SURVEY.md section 8c: reference functions examples/subgrid/kernels.inl:1-332, g++ -O0 -ffp-contract=off",
    "uL": [2, -1, 0.1, 0, 6.5],
    "uR": [1, 0.5, 0.05, 0, 6.4],
    "kepes_f64": [-0.0033677849363779044, 1.3325002339044238, -0.00016838924681889661, 0, -0.020038728337171019],
    "kepes_f32": [-0.00336763263, 1.33249998, -0.000168381259, 0, -0.0200378895],
    "hll_f32": [0.682511926, 1.34832394, 0.0341256, 0, -0.507861376],
}
if __name__ == "__main__":
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "survey_8c_kat.json"), "w") as f:
        json.dump(KAT, f, indent=1)
