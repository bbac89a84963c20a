"""RNNoise-nu text model -> nnnoiseless binary .rnn (what the reference's train/convert_rnnoise.py does:
drop the header line, write every integer modulo 256 as one byte).
Usage: python make_sh_rnn.py /root/reference/test_data/sh.rnnn sh.rnn"""
import sys


def convert(text: str) -> bytes:
    head, _, body = text.partition("\n")
    if head.strip() != "rnnoise-nu model file version 1":
        raise ValueError("unexpected model header")
    return bytes(int(tok) % 256 for tok in body.split())


if __name__ == "__main__":
    open(sys.argv[2], "wb").write(convert(open(sys.argv[1]).read()))
