"""`python -m speech_recognition_amd.run.evaluate` - speech_recognition/run/evaluate.py on MI355X: decode a
dataset (greedy search, or beam search with --beam-size) with a trained model, report WER / CER, optionally write
a (Prediction, Target, WER, CER) TSV.  Same flags as the reference.  --device
must be GPU; --mixed-precision selects bf16 operands for the dense contractions."""
import argparse
import csv
import sys

from ..configs import DataConfig
from ..data import SentencePieceTokenizer, get_dataset, get_tfrecord_dataset
from ..utils import get_device_strategy, get_logger, levenshtein_distance
from ._decode import feature_fn, load_model_and_searcher, strip_tokens

# fmt: off
parser = argparse.ArgumentParser("This is script to inferece (generate sentence) with seq2seq model")
parser.add_argument("--data-config", type=str, required=True, help="data processing config file")
parser.add_argument("--model-config", type=str, required=True, help="model config file")
parser.add_argument("--dataset-paths", required=True, help="a tsv/tfrecord dataset file or multiple files ex) *.tsv")
parser.add_argument("--model-path", type=str, required=True, help="pretrained model checkpoint")
parser.add_argument("--sp-model-path", type=str, required=True, help="sentencepiece model path")
parser.add_argument("--output-path", help="output tsv file path to save generated sentences")
parser.add_argument("--batch-size", type=int, default=512)
parser.add_argument("--beam-size", type=int, default=0, help="not given, use greedy search else beam search with this value as beam size")
parser.add_argument("--use-tfrecord", action="store_true", help="use tfrecord dataset")
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

    logger.info(f"[+] Load Tokenizer from {args.sp_model_path}")
    tokenizer = SentencePieceTokenizer(args.sp_model_path, add_bos=True, add_eos=True)
    bos_id, eos_id = tokenizer.tokenize("").tolist()
    logger.info(f"[+] Load Data Config from {args.data_config}")
    config = DataConfig.from_yaml(args.data_config)

    if args.use_tfrecord:
        logger.info(f"[+] Load TFRecord dataset from {args.dataset_paths}")
        dataset = get_tfrecord_dataset(args.dataset_paths)
    else:
        logger.info(f"[+] Load dataset from {args.dataset_paths}")
        dataset = get_dataset(args.dataset_paths, config.file_format, config.sample_rate, tokenizer)
    if config.use_delta_accelerate:
        logger.info("[+] Use delta and deltas accelerate")

    model, searcher = load_model_and_searcher(config, args.model_config, args.model_path, tokenizer, logger)
    features = feature_fn(config, args.use_tfrecord)
    dataset = dataset.padded_batch(args.batch_size, with_lengths=True)

    logger.info("[+] Start Inference")
    outputs = []
    for (audio, target), (n_audio, _) in dataset:
        if args.beam_size > 0:
            tokens = searcher.beam_search(features(audio, n_audio), args.beam_size)[0][:, 0, :].cpu().numpy()
        else:
            tokens = searcher.greedy_search(features(audio, n_audio))[0].cpu().numpy()
        outputs.extend(zip(tokens, target))
    logger.info("[+] Ended Inference")

    to_str = lambda row: tokenizer.detokenize(strip_tokens(row, bos_id, eos_id))
    outputs = [(to_str(pred), to_str(target)) for pred, target in outputs]

    wers, cers = [], []
    for pred, target in outputs:
        wers.append(levenshtein_distance(target.split(), pred.split(), True) if target.split() else float(bool(pred.split())))
        cers.append(levenshtein_distance(target, pred, True) if target else float(bool(pred)))
    logger.info(f"[+] Average WER: {sum(wers) / max(len(wers), 1) * 100:.4f}%")
    logger.info(f"[+] Average CER: {sum(cers) / max(len(cers), 1) * 100:.4f}%")

    if args.output_path:
        with open(args.output_path, "w", newline="") as fout:
            wtr = csv.writer(fout, delimiter="\t")
            wtr.writerow(["Prediction", "Target", "WER", "CER"])
            for (pred, target), wer, cer in zip(outputs, wers, cers):
                wtr.writerow((pred, target, wer, cer))
        logger.info(f"[+] Saved (Prediction, Target) pairs to {args.output_path}")


if __name__ == "__main__":
    sys.exit(main(parser.parse_args()))
