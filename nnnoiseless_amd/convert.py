"""RNNoise text model -> nnnoiseless binary .rnn, the reference's train/convert_rnnoise.py as a library + CLI.

    python -m nnnoiseless_amd.convert INPUT OUTPUT

Pure host-side text processing (no GPU, no native library needed); the C ABI offers the same as
nnn_convert_rnnoise_text / nnn_model_from_rnnoise_text for C hosts.
"""
import sys

HEADER = "rnnoise-nu model file version 1"


def convert_rnnoise_text(text):
    """train/convert_rnnoise.py:18-29: first line must be the header; every following integer modulo 256 is a byte."""
    if isinstance(text, bytes):
        text = text.decode()
    head, _, body = text.partition("\n")
    if head.strip() != HEADER:
        raise ValueError("Unexpected input file format")
    return bytes(int(tok) % 256 for tok in body.split())


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 3:                                   # train/convert_rnnoise.py:10-13
        print("Expected two arguments.")
        print("USAGE: convert_rnnoise.py INPUT OUTPUT")
        return 1
    with open(argv[1], "r") as f:
        try:
            data = convert_rnnoise_text(f.read())
        except ValueError as e:
            print(e)
            return 1
    with open(argv[2], "wb") as f:
        f.write(data)
    print(f"Converted {argv[1]} to {argv[2]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
