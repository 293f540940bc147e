"""`python -m speech_recognition_amd.run.make_tfrecord` - speech_recognition/run/make_tfrecord.py on MI355X:
TSV datasets -> GZIP TFRecord files of (audio feature tensor, token tensor) examples, one .tfrecord per
.tsv.  The features come from the same GPU kernel the training step uses (DataConfig.audio_feature_fn), so
training from the records gives the same inputs as training from the audio files; the files are readable
by the reference's get_tfrecord_dataset (format check: tests/test_data_host.py reproduces the reference's
fixture byte for byte)."""
import argparse
import glob
import os
import sys

from ..configs import DataConfig
from ..data import SentencePieceTokenizer, get_dataset
from ..tfrecord import TFRecordWriter
from ..utils import get_logger

# fmt: off
parser = argparse.ArgumentParser()
parser.add_argument("--data-config", type=str, required=True, help="data processing config file")
parser.add_argument("--dataset-paths", type=str, required=True, help="dataset file path glob pattern")
parser.add_argument("--output-dir", type=str, help="output directory path, default is input dataset file directoruy")
parser.add_argument("--sp-model-path", type=str, default="resources/sp-model/sp_model_unigram_16K.model", help="sentencepiece model path")
# fmt: on


def main(args: argparse.Namespace):
    logger = get_logger("make-tfrecord")
    input_files = sorted(glob.glob(args.dataset_paths))
    logger.info(f"[+] Number of Dataset Files: {len(input_files)}")
    logger.info(f"[+] Load Config From {args.data_config}")
    config = DataConfig.from_yaml(args.data_config)
    logger.info(f"[+] Load Tokenizer From {args.sp_model_path}")
    tokenizer = SentencePieceTokenizer(args.sp_model_path, add_bos=True, add_eos=True)
    feature_fn = config.audio_feature_fn

    logger.info("[+] Start Saving Dataset...")
    for file_path in input_files:
        output_dir = args.output_dir if args.output_dir else os.path.dirname(file_path)
        os.makedirs(output_dir or ".", exist_ok=True)
        output_path = os.path.join(output_dir, os.path.splitext(os.path.basename(file_path))[0] + ".tfrecord")
        count = 0
        with TFRecordWriter(output_path) as writer:
            for audio, tokens in get_dataset(file_path, config.file_format, config.sample_rate, tokenizer).prefetch(8):
                writer.write(feature_fn(audio).cpu().numpy(), tokens)
                count += 1
        logger.info(f"    {file_path} -> {output_path} ({count} examples)")
    logger.info("[+] Done")


if __name__ == "__main__":
    sys.exit(main(parser.parse_args()))
