"""`python -m speech_recognition_amd.run.inference` - speech_recognition/run/inference.py on MI355X: decode
audio files (greedy search, or beam search with --beam-size) and write an (AudioPath, DecodedSentence) TSV.  Same
flags as the reference; --device must be GPU, --mixed-precision selects bf16 operands for the dense contractions."""
import argparse
import csv
import glob
import sys

import numpy as np

from ..configs import DataConfig
from ..data import Dataset, SentencePieceTokenizer, load_audio_file
from ..utils import get_device_strategy, get_logger
from ._decode import feature_fn, load_model_and_searcher, strip_tokens

# fmt: off
parser = argparse.ArgumentParser("This is script to inferece (generate sentence) with seq2seq model")
parser.add_argument("--data-config", type=str, required=True, help="data processing config file")
parser.add_argument("--model-config", type=str, required=True, help="model config file")
parser.add_argument("--audio-files", required=True, help="an audio file or glob pattern of multiple files ex) *.pcm")
parser.add_argument("--model-path", type=str, required=True, help="pretrained model checkpoint")
parser.add_argument("--output-path", default="output.tsv", help="output tsv file path to save generated sentences")
parser.add_argument("--sp-model-path", type=str, required=True, help="sentencepiece model path")
parser.add_argument("--batch-size", type=int, default=512)
parser.add_argument("--beam-size", type=int, default=0, help="not given, use greedy search else beam search with this value as beam size")
parser.add_argument("--mixed-precision", action="store_true", help="Use mixed precision FP16")
parser.add_argument("--device", type=str, default="CPU", help="device to train model")
# fmt: on


def main(args: argparse.Namespace):
    get_device_strategy(args.device)
    logger = get_logger("inference")
    if args.mixed_precision:
        from .. import ops
        ops.set_mixed_precision(True)
        logger.info("[+] --mixed-precision: dense contractions use bf16 operands with f32 accumulation")

    tokenizer = SentencePieceTokenizer(args.sp_model_path, add_bos=True, add_eos=True)
    bos_id, eos_id = tokenizer.tokenize("").tolist()
    dataset_files = sorted(glob.glob(args.audio_files))
    if not dataset_files:
        logger.error("[Error] Dataset path is invalid!")
        sys.exit(1)

    logger.info(f"Load Data Config from {args.data_config}")
    config = DataConfig.from_yaml(args.data_config)
    load = load_audio_file(config.sample_rate, config.file_format, config.sample_rate)
    dataset = Dataset(lambda: ((load(path), np.zeros(0, np.int32)) for path in dataset_files))
    if config.use_delta_accelerate:
        logger.info("[+] Use delta and deltas accelerate")
    dataset = dataset.padded_batch(args.batch_size, with_lengths=True).prefetch(2)

    model, searcher = load_model_and_searcher(config, args.model_config, args.model_path, tokenizer, logger)
    features = feature_fn(config, stored_features=False)

    logger.info("Start Inference")
    outputs = []
    for (audio, _), (n_audio, _) in dataset:
        if args.beam_size > 0:
            outputs.extend(searcher.beam_search(features(audio, n_audio), args.beam_size)[0][:, 0, :].cpu().numpy())
        else:
            outputs.extend(searcher.greedy_search(features(audio, n_audio))[0].cpu().numpy())
    outputs = [tokenizer.detokenize(strip_tokens(row, bos_id, eos_id)) for row in outputs]
    logger.info("Ended Inference, Start to save...")

    with open(args.output_path, "w", newline="") as fout:
        wtr = csv.writer(fout, delimiter="\t")
        wtr.writerow(["AudioPath", "DecodedSentence"])
        for audio_path, decoded_sentence in zip(dataset_files, outputs):
            wtr.writerow((audio_path, decoded_sentence))
    logger.info(f"Saved (AudioPath, DecodedSentence) pairs to {args.output_path}")


if __name__ == "__main__":
    sys.exit(main(parser.parse_args()))
