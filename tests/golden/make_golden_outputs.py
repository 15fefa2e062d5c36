"""Mint golden vectors for the tabular outputs from the REFERENCE ITSELF (build container only):
``classpose.entrypoints.outputs.calculate_cellular_densities`` (pure python + pandas) and
``predict_wsi.get_artefact_class_id``, imported under the stub finder of make_golden.py.
Fixtures hold inputs and outputs only.   python tests/golden/make_golden_outputs.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REF)
    from classpose.entrypoints.outputs import calculate_cellular_densities
    import classpose.entrypoints.predict_wsi as pw
    rng = np.random.default_rng(11)
    labels = ["Neutrophil", "Epithelial", "Lymphocyte", "Plasma cell", "Eosinophil", "Connective"]

    def cells(n):
        names = labels + ["Other"]                              # one class outside the label list
        return [{"properties": {"classification": {"name": names[int(k)]}}} for k in rng.integers(0, len(names), n)]
    cases = []
    for tissue, art, mx, my, n in ((3.0e6, 1.2e5, 0.5, 0.5, 400), (1.0e5, 2.0e5, 0.25, 0.26, 50), (7.7e7, 0, 0.22, 0.22, 0)):
        c = cells(n)
        df = calculate_cellular_densities(c, tissue, art, mx, my, labels)
        cases.append(dict(cells=c, tissue=tissue, artefact=art, mpp_x=mx, mpp_y=my, rows=df.to_dict("records")))
    by_roi = {"Tumour": cells(120), "Stroma": cells(30), "Empty": []}
    df = calculate_cellular_densities(by_roi, {"Tumour": 4.0e5, "Stroma": 9.0e4}, {"Tumour": 1.0e4, "Stroma": 9.5e4},
                                      0.5, 0.5, labels)
    roi_case = dict(cells=by_roi, tissue={"Tumour": 4.0e5, "Stroma": 9.0e4}, artefact={"Tumour": 1.0e4, "Stroma": 9.5e4},
                    mpp_x=0.5, mpp_y=0.5, rows=df.to_dict("records"))
    ids = {n: pw.get_artefact_class_id(n) for n in ("Fold", "Darkspot & Foreign Object", "PenMarking",
                                                    "Edge & Air Bubble", "OOF", "Normal Tissue", "nonsense")}
    with open(os.path.join(HERE, "reference_outputs.json"), "w") as f:
        json.dump(dict(labels=labels, global_cases=cases, roi_case=roi_case, artefact_class_ids=ids), f)
    print("wrote reference_outputs.json")


if __name__ == "__main__":
    main()
