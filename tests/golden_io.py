"""Loads the committed golden vectors (tests/golden/*.npz: inputs + outputs produced by the reference)."""
import os

import numpy as np

import oracle_lib as ol

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PICTURE_CASES = ["base_p_8b", "base_b_8b", "base_p_10b", "main_b_10b", "main_admvp_only", "main_iqt_only",
                 "main_addb_10b", "main_addb_8b_shared_refs", "main_alf_10b", "main_alf_8b_across_tiles", "main_alf_only_luma",
                 "main_ctu128_10b", "main_ctu128_8b_noiqt", "main_ats_10b", "main_ats_8b_noiqt",
                 "main_atsinter_10b", "main_atsinter_8b_mixed", "main_atsinter_noaddb",
                 "main_btt_10b", "main_btt_ctu128_8b", "main_btt_noaddb_8b", "main_ctu128_noaddb_8b",
                 "base_i_8b", "base_p_constrained_intra_10b", "main_i_btt_10b", "main_b_ctu128_intra_mix_8b",
                 "main_eipd_i_10b", "main_eipd_i_btt_8b", "main_eipd_b_ctu128_constrained_10b",
                 "main_affine_b_10b", "main_affine_p_8b_atsinter", "main_affine_b_ctu128_10b",
                 "main_ibc_i_10b", "main_ibc_b_8b_noaddb", "main_ibc_p_ctu128_eipd_10b",
                 "main_htdf_b_10b", "main_htdf_i_8b_constrained", "main_htdf_p_ctu128_10b", "main_dmvr_b_10b", "main_dmvr_b_8b_ctu128_mixed",
                 "base_p_12b", "base_b_12b", "main_addb_alf_12b", "main_all_tools_b_12b", "main_i_eipd_ibc_htdf_ctu128_12b"]


def load_picture_case(name):
    d = np.load(os.path.join(GOLDEN, f"pic_{name}.npz"))
    w, h, bd, admvp, iqt = (int(v) for v in d["params"])
    refs = {}
    for l in range(2):
        for i in range(int(d["n_refs"][l])):
            if f"refalias_{i}_{l}" in d.files:
                continue
            pic = ol.Picture(w, h, int(d[f"refpoc_{i}_{l}"]), [d[f"ref_{i}_{l}_{c}"] for c in range(3)])
            pic.pad_numpy()
            refs[(i, l)] = pic
    for l in range(2):
        for i in range(int(d["n_refs"][l])):
            if f"refalias_{i}_{l}" in d.files:
                refs[(i, l)] = refs[tuple(int(v) for v in d[f"refalias_{i}_{l}"])]
    tools = [int(v) for v in d["tools"]] if "tools" in d.files else []
    tools = (tools + [0, 0, 0, 0, 0, 6, 0][len(tools):])[:7]
    alf_params = None
    if "alf_enable" in d.files:
        alf_params = {"enable": tuple(int(v) for v in d["alf_enable"]), "luma_coef": d["alf_luma_coef"], "chroma_coef": d["alf_chroma_coef"],
                      "ctb_flag": d["alf_ctb_flag"], "across_tiles": int(d["alf_across_tiles"])}
    batch = {k[2:]: d[k] for k in d.files if k.startswith("b_")}
    batch["n_coef"] = int(batch["n_coef"])
    batch["constrained_intra_pred"] = int(batch.get("constrained_intra_pred", 0))
    batch["htdf_slice_qp"] = int(batch.get("htdf_slice_qp", 0))
    batch.setdefault("cbf_sub", None)
    batch.setdefault("ats", None)
    batch.setdefault("ats_inter", None)
    case = {"name": name, "w": w, "h": h, "bd": bd, "admvp": admvp, "iqt": iqt, "refs": refs, "batch": batch,
            "addb": tools[0], "alf": tools[1], "alpha_off": tools[2], "beta_off": tools[3], "no_deblock": tools[4], "log2_ctu": tools[5],
            "eipd": tools[6], "alf_params": alf_params}
    expect = {"out": [d[f"out_{c}"] for c in range(3)], "pre": [d[f"pre_{c}"] for c in range(3)], "resid": d["resid"],
              "map_scu": d["map_scu"], "dmvr_mv": d["dmvr_mv"] if "dmvr_mv" in d.files else None}
    return case, expect
