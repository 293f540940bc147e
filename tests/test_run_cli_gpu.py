"""run.evaluate / run.inference / run.make_tfrecord on the GPU - the reference's tests/run/test_evaluate.py,
test_inference.py and test_make_tfrecord.py restated on its own fixtures (mini model configs, the checkpoints
TensorFlow wrote for them, the two-clip dataset), greedy search only."""
import csv
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "reference_fixtures")
LIBRI = os.path.join(ROOT, "resources", "configs", "libri_config.yml")
SP_MODEL = os.path.join(FIX, "sp_model_unigram_16K_libri.model")
MODELS = [(os.path.join(FIX, "las_mini_for_test.yml"), os.path.join(FIX, "las.ckpt")),
          (os.path.join(FIX, "deepspeech_mini_for_test.yml"), os.path.join(FIX, "ds.ckpt"))]
WAV = os.path.join(FIX, "wav_dataset.tsv")
TFREC = os.path.join(FIX, "wav_dataset.tfrecord")


@pytest.mark.parametrize("use_tfrecord", [False, True])
@pytest.mark.parametrize("mixed_precision", [False, True])
@pytest.mark.parametrize("model", MODELS)
def test_evaluate(tmp_path, model, mixed_precision, use_tfrecord):
    from speech_recognition_amd.run.evaluate import main, parser
    out = tmp_path / "eval.tsv"
    args = ["--data-config", LIBRI, "--model-config", model[0], "--dataset-paths", TFREC if use_tfrecord else WAV, "--model-path", model[1],
            "--output-path", str(out), "--sp-model-path", SP_MODEL, "--batch-size", "4", "--device", "GPU"]
    if mixed_precision:
        args.append("--mixed-precision")
    if use_tfrecord:
        args.append("--use-tfrecord")
    assert main(parser.parse_args(args)) is None
    rows = list(csv.reader(open(out), delimiter="\t"))
    assert rows[0] == ["Prediction", "Target", "WER", "CER"] and len(rows) == 3        # header + the two clips
    for row in rows[1:]:
        assert np.isfinite(float(row[2])) and np.isfinite(float(row[3]))
    if not use_tfrecord:
        assert rows[1][1] != ""                                  # the target sentence survives tokenise -> detokenise
    # --beam-size (run/evaluate.py:97-99): the best hypothesis of the beam is scored instead of the greedy one
    assert main(parser.parse_args(args + ["--beam-size", "2"])) is None
    rows = list(csv.reader(open(out), delimiter="\t"))
    assert len(rows) == 3 and all(np.isfinite(float(r[2])) for r in rows[1:])


@pytest.mark.parametrize("model", MODELS)
def test_inference(tmp_path, model):
    from speech_recognition_amd.run.inference import main, parser
    out = tmp_path / "out.tsv"
    audio = os.path.join(FIX, "audio_files", "test.flac")
    args = ["--data-config", LIBRI, "--model-config", model[0], "--audio-files", audio, "--model-path", model[1], "--output-path", str(out),
            "--sp-model-path", SP_MODEL, "--batch-size", "4", "--device", "GPU"]
    assert main(parser.parse_args(args)) is None
    rows = list(csv.reader(open(out), delimiter="\t"))
    assert rows[0] == ["AudioPath", "DecodedSentence"] and len(rows) == 2 and rows[1][0] == audio
    assert main(parser.parse_args(args + ["--beam-size", "3"])) is None
    rows = list(csv.reader(open(out), delimiter="\t"))
    assert rows[0] == ["AudioPath", "DecodedSentence"] and len(rows) == 2 and rows[1][0] == audio
    with pytest.raises(SystemExit):
        main(parser.parse_args(args[:5] + [str(tmp_path / "none*.wav")] + args[6:]))


def test_make_tfrecord_matches_reference_features(tmp_path):
    from speech_recognition_amd.data import SentencePieceTokenizer, get_tfrecord_dataset
    from speech_recognition_amd.run.make_tfrecord import main, parser
    args = ["--data-config", LIBRI, "--dataset-paths", WAV, "--sp-model-path", SP_MODEL, "--output-dir", str(tmp_path)]
    assert main(parser.parse_args(args)) is None
    out = os.path.join(tmp_path, "wav_dataset.tfrecord")
    assert os.path.exists(out)                                                          # tests/run/test_make_tfrecord.py
    mine, ref = list(get_tfrecord_dataset(out)), list(get_tfrecord_dataset(TFREC))
    assert len(mine) == len(ref) == 2
    tok = SentencePieceTokenizer(SP_MODEL)
    for (feat, tokens), (ref_feat, _), text in zip(mine, ref, ["Hello World Good night", "gOddy bye"]):
        np.testing.assert_array_equal(feat, ref_feat)           # the reference's own log-mel of the same clips, bit for bit
        assert tokens.tolist() == tok.tokenize(text).tolist()   # (the fixture's tokens came from a character tokenizer)

