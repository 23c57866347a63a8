"""Hand-off formats, host logic only: the reference's own checked-in outputs
(tests/golden/handoff/, see its README) are reproduced key for key, and byte for byte where the bytes do not
depend on audio."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN

H = os.path.join(GOLDEN, "handoff")


def _load(name):
    with open(os.path.join(H, name)) as fp:
        text = fp.read()
    return text, json.loads(text)


def test_json_formatting_is_the_references():
    from sos_amd import handoff
    for name in ("m1_eval_results.json", "m1_pred_data.json"):
        text, obj = _load(name)
        assert json.dumps(obj, **handoff.JSON_DUMP_PARAMS) == text


def test_show_metrics_reproduces_the_golden_statistics():
    from sos_amd import handoff
    _, ev = _load("m1_eval_results.json")
    labels = [b for it in ev["data"] for b in it["label"]]
    preds = [b for it in ev["data"] for b in it["pred_label"]]
    got = handoff.show_metrics(labels, preds)
    assert list(got.items()) == list(ev["prediction_statistics"]["all"].items())     # keys, order, values (incl. null)
    # a case with both classes and every rate defined (hand-computed)
    m = handoff.show_metrics([0, 0, 1, 1, 1, 0], [0, 1, 1, 0, 1, 0])
    assert (m["true_positive"], m["false_positive"], m["true_negative"], m["false_negative"]) == (2, 1, 2, 1)
    assert m["num_silent_samples"] == 3 and m["base"] == 0.5 and abs(m["accuracy"] - 4 / 6) < 1e-15
    assert abs(m["f1"] - 4 / 6) < 1e-15 and abs(m["mcc"] - 1 / 3) < 1e-12 and abs(m["roc_auc"] - 2 / 3) < 1e-15


def test_pred_data_from_eval_results_matches_golden(tmp_path):
    from sos_amd import handoff
    shutil.copy(os.path.join(H, "m1_eval_results.json"), tmp_path / "eval_results.json")
    out = handoff.create_data_from_prediction(str(tmp_path / "eval_results.json"), save_results=False)
    assert os.path.basename(out) == "pred_data.json"
    with open(out) as fp:
        got_text = fp.read()
    _, want = _load("m1_pred_data.json")
    for f in want["files"]:
        assert f.pop("mixed_audio").startswith("recovered/")          # written only with save_results (needs the audio)
    assert got_text == json.dumps(want, **handoff.JSON_DUMP_PARAMS)   # byte for byte
    assert handoff.create_data_from_prediction(str(tmp_path / "eval_results.json"), suffix="_0_5", noise_snr=2.5,
                                               save_results=False).endswith("pred_data_0_5_snr2_5.json")


def test_suffix_helpers_and_bitstream_trim():
    from sos_amd import handoff
    assert handoff.convert_snr_to_suffix2(None) == "" and handoff.convert_snr_to_suffix2(10.0) == "_snr10"
    assert handoff.convert_snr_to_suffix2(2.5) == "_snr2_5" and handoff.convert_snr_to_suffix2("x") == ""
    assert handoff.convert_threshold_to_suffix("0.5") == "_0_5" and handoff.convert_threshold_to_suffix("") == ""
    assert handoff.convert_threshold_to_suffix("1.5") == ""
    assert handoff._trim_unknown("2221101222") == (3, 7) and handoff._trim_unknown("1101") == (0, 4)
    assert handoff.find_common_path("/a/b/c1/x.wav", "/a/b/c2/y.wav") == "/a/b"


def test_dataset_json_schema_fields_used():
    _, ds = _load("dataset_sounds_of_silence.json")
    for f in ds["files"]:
        for k in ("path", "audio_path", "bit_stream", "num_frames", "framerate", "audio_sample_rate", "audio_samples", "duration"):
            assert k in f
        assert len(f["bit_stream"]) == f["num_frames"]
