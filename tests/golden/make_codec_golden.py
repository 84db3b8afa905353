"""Writes codec_vectors.json: byte strings compressed with the C libraries of the Snappy, LZ4 and Zstandard
formats (through pyarrow.Codec, which links the same libraries snappy-java and zstd-jni wrap), for the page
decompression tests.  Each vector: codec, the compressed bytes, and size and SHA-256 of the original, which
tests/codec_inputs.py regenerates.  Run by hand:  python tests/golden/make_codec_golden.py"""
import base64
import hashlib
import json
import os

import sys

import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from codec_inputs import inputs  # noqa: E402


def main():
    vectors = []
    for name, raw in inputs():
        for codec in ("snappy", "lz4_raw", "zstd"):
            levels = (1, 9) if codec == "zstd" else (None,)
            for level in levels:
                c = pa.Codec(codec, compression_level=level) if level else pa.Codec(codec)
                comp = c.compress(raw, asbytes=True)
                assert pa.Codec(codec).decompress(comp, len(raw), asbytes=True) == raw
                vectors.append(dict(name=name + ("_l%d" % level if level else ""), codec=codec.upper(), size=len(raw),
                                    sha256=hashlib.sha256(raw).hexdigest(),
                                    compressed=base64.b64encode(comp).decode()))
                print(name, codec, level, len(raw), "->", len(comp))
    with open(os.path.join(HERE, "codec_vectors.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_codec_golden.py", pyarrow=pa.__version__, vectors=vectors), f)


if __name__ == "__main__":
    main()
