#!/usr/bin/env python3
"""Convert the JPEG images of a data set to PNG, in place or into another directory, and rewrite its image list.

The reference's DataSetCam reads JPEG through libgd (src/VideoLib/datasetcam.cpp:128-131, gdImageCreateFromJpeg = libjpeg);
the host library of this repository decodes PNG / PGM / PPM and baseline and progressive JPEG itself (rebvo_amd/host/src/png_reader.cpp,
jpeg_reader.cpp) — EuRoC and TUM, the data sets of every BASELINE configuration, ship PNG.  For what the JPEG reader does not
take (arithmetic-coded, CMYK, 12-bit files), decode once with this tool (PIL = libjpeg(-turbo), the decoder family libgd uses) and point DataSetDir / DataSetFile at the result:

    tools/jpeg_to_png.py <DataSetDir> <DataSetFile> <out_dir>

writes <out_dir>/<name>.png for every listed <name>.jpg / .jpeg and <out_dir>/list.txt with the same time stamps.
"""
import os
import sys


def main():
    if len(sys.argv) != 4:
        print(__doc__)
        return 2
    from PIL import Image
    ddir, dfile, out = sys.argv[1:]
    os.makedirs(out, exist_ok=True)
    n = 0
    with open(dfile) as f, open(os.path.join(out, "list.txt"), "w") as lst:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            sep = "," if "," in line else " "
            stamp, name = [v.strip() for v in line.split(sep, 1)]
            base, ext = os.path.splitext(name)
            dst = base.replace("/", "_") + ".png"
            if ext.lower() in (".jpg", ".jpeg"):
                Image.open(os.path.join(ddir, name)).convert("RGB").save(os.path.join(out, dst))
            else:
                Image.open(os.path.join(ddir, name)).save(os.path.join(out, dst))
            lst.write(f"{stamp}{sep}{dst}\n")
            n += 1
    print(f"{n} images -> {out} (DataSetDir={out}/ DataSetFile={os.path.join(out, 'list.txt')})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
