#!/usr/bin/env python3
"""python gen_cfeatures.py --dir D [--after YYYY-MM-DD]      (same flags as the reference, gen_cfeatures.py:462-480)

Builds the character-feature index: every image under D is resized to 384x384 (bilinear) and CLIP-
normalised on 8 host threads (gen_cfeatures.py:100-110,285-295), encoded in batches by the device CCIP
encoder (hiptagsearch.cfeatures.CCIPEncoder, replacing the onnxruntime session of :112-118,158), the
path is appended to charactor-featues-idx.csv (:376) and the unit-normalised feature row to the device
index `charactor-featues-idx` (:307-315), saved at the end (:459).  With --after, only files modified on
or after the date are encoded and appended to the existing index (the reference copies the old index into
a new revision first, :340-368; here the index file is loaded and extended in place, a .bak copy is kept).

Extra switches: --checkpoint ccip.safetensors (timm MetaFormer key layout; without it the seeded synthetic
stand-in is used -- there is no network here to fetch deepghs/ccip_onnx), --batch, --device."""
import argparse
import concurrent.futures
import datetime
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

INDEX_PREFIX = 'charactor-featues-idx'
EXTENSIONS = ('.png', '.jpg', '.jpeg', '.gif', '.webp', '.bmp', '.PNG', '.JPG', '.JPEG', '.GIF', '.WEBP', '.BMP')   # :44-45
WORKER_NUM = 8                                                                                                     # :49


def list_files_recursive(dir_path: str):
    out = []
    for root, _, files in os.walk(dir_path):
        for f in files:
            if os.path.splitext(f)[1] in EXTENSIONS:
                out.append(os.path.join(root, f))
    return out


def main(arg_str: list) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--dir', nargs=1, required=True, help='tagging target directory path')
    parser.add_argument('--after', nargs=1, help='tagging new images after this date (mtime attribute). Format: YYYY-MM-DD')
    parser.add_argument('--checkpoint', default=None)
    parser.add_argument('--batch', type=int, default=64)
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--workers', type=int, default=0,
                        help='decode / resize in this many processes (hiptagsearch/pipeline.py); the uint8 images go to the device u8 entry point')
    parser.add_argument('--operands', choices=['bf16', 'half', 'e4m3'], default='bf16',
                        help='MFMA operand type of the encoder GEMMs (e4m3 = the fp8 mode: faster, 3 mantissa bits)')
    args = parser.parse_args(arg_str)
    after_date = None
    if args.after is not None:
        try:
            after_date = datetime.datetime.strptime(args.after[0], '%Y-%m-%d').date()
        except Exception as e:
            print('%s: %s' % (type(e), str(e)))
            print('Invalid date format. format is YYYY-MM-DD')
            raise SystemExit(1)

    import numpy as np
    from hiptagsearch import synth
    from hiptagsearch.cfeatures import CCIPEncoder, CharacterFeatureIndex, gen_image_ndarray
    from hiptagsearch.index import Similarity
    cfg = dict(synth.CCIP_B36_384, operand_f16={'bf16': 0, 'half': 1, 'e4m3': 2}[args.operands])
    if args.checkpoint:
        encoder = CCIPEncoder.from_safetensors(args.checkpoint, cfg, max_batch=args.batch, device=args.device)
    else:
        print('no --checkpoint: using the seeded synthetic CCIP weights')
        encoder = CCIPEncoder(cfg, synth.ccip_weights(cfg), max_batch=args.batch, device=args.device)
    cindex = CharacterFeatureIndex(encoder, device=args.device, prefix=INDEX_PREFIX)

    file_list = list_files_recursive(args.dir[0])
    print(f'{len(file_list)} files found')
    if after_date is not None:
        file_list = [p for p in file_list if datetime.date.fromtimestamp(os.path.getmtime(p)) >= after_date]
        print(f'{len(file_list)} files found after {after_date}')
        if os.path.exists(INDEX_PREFIX):
            for f in (INDEX_PREFIX, INDEX_PREFIX + '.npy', INDEX_PREFIX + '.csv'):
                if os.path.exists(f):
                    shutil.copy2(f, f + '.bak')
            cindex.index = Similarity.load(INDEX_PREFIX, device=args.device)
            if os.path.exists(INDEX_PREFIX + '.csv'):
                cindex.paths = [l.rstrip('\n') for l in open(INDEX_PREFIX + '.csv', encoding='utf-8')]

    start = time.perf_counter()
    done = 0
    if args.workers > 0:
        from hiptagsearch import pipeline
        with open(INDEX_PREFIX + '.csv', 'a', encoding='utf-8') as fcsv, \
                pipeline.DecodePool(args.workers, cfg["image_size"], args.batch, pipeline.CCIP) as dpool:
            for kept, images in dpool.batches(file_list):
                cindex.add_features(kept, encoder.forward_u8(images))                # /255 and the CLIP normalisation on the device
                for p in kept:
                    fcsv.write(p + '\n')
                done += len(kept)
                el = time.perf_counter() - start
                print(f'{done} files processed\n{el:.2f} seconds elapsed\n{el / max(done, 1):.4f} seconds per file\n', flush=True)
        cindex.index.save(INDEX_PREFIX)
        return
    with open(INDEX_PREFIX + '.csv', 'a', encoding='utf-8') as fcsv, \
            concurrent.futures.ThreadPoolExecutor(max_workers=WORKER_NUM) as pool:
        nxt = pool.map(gen_image_ndarray, file_list[:args.batch])                    # one batch of decode ahead of the device
        for s in range(0, len(file_list), args.batch):
            arrs = list(nxt)
            if s + args.batch < len(file_list):
                nxt = pool.map(gen_image_ndarray, file_list[s + args.batch: s + 2 * args.batch])
            keep = [(p, a) for p, a in zip(file_list[s:s + args.batch], arrs) if a is not None]      # failed loads are skipped (:392-395)
            if not keep:
                continue
            feats = cindex.ccip_batch_extract_features([a for _, a in keep])
            cindex.add_features([p for p, _ in keep], feats)
            for p, _ in keep:
                fcsv.write(p + '\n')                                                 # :376
            done += len(keep)
            el = time.perf_counter() - start
            print(f'{done} files processed\n{el:.2f} seconds elapsed\n{el / max(done, 1):.4f} seconds per file\n', flush=True)
    cindex.index.save(INDEX_PREFIX)


if __name__ == "__main__":
    main(sys.argv[1:])
