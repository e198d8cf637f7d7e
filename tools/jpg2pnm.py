#!/usr/bin/env python3
"""Convert <dense_folder>/images/*.jpg to binary PGM (grey, what APD::InuputInitialization reads)
and PPM (colour, what RunFusion reads): the host library has no JPEG decoder (no OpenCV/libjpeg)."""
import glob
import os
import sys

from PIL import Image

for p in sorted(glob.glob(os.path.join(sys.argv[1], "images", "*.jpg"))):
    im = Image.open(p)
    im.convert("L").save(p[:-4] + ".pgm")
    im.convert("RGB").save(p[:-4] + ".ppm")
    print(p)
