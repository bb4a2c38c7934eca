#!/usr/bin/env python3
"""python gen_cfeatures.py --dir D [--after YYYY-MM-DD]      (same flags as the reference, gen_cfeatures.py:462-480)

Builds the character-feature index: every image under D is resized to 384x384 (bilinear) and CLIP-
normalised on 8 host threads (gen_cfeatures.py:100-110,285-295), encoded in batches by the device CCIP
encoder (hiptagsearch.cfeatures.CCIPEncoder, replacing the onnxruntime session of :112-118,158), the
path is appended to charactor-featues-idx.csv (:376) and the unit-normalised feature row to the device
index `charactor-featues-idx` (:307-315), saved at the end (:459).

--after (gen_cfeatures.py:340-370): every `charactor-featues-idx*` file is first copied into a directory named
YYYYmmdd_HHMMSS, the latest revision N (`charactor-featues-idx` = 0, `charactor-featues-idxN`) is loaded and copied into
the new revision `charactor-featues-idx<N+1>` (in one block here, vector by vector in the reference), the
new features are appended and THAT revision is saved; the query side loads the highest revision (webui.py:272-277,
cfeatures.CharacterFeatureIndex.load_latest).  The paths csv is shared by all revisions and appended in place.

Multi-GPU (one process per GPU): under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P gen_cfeatures.py --dir D
rank r encodes a contiguous block of the file list, ONE all-gather of the float32[768] rows puts them in file order,
rank 0 owns the index and the csv: the files written are those of the single-process run.

Extra switches: --checkpoint ccip.safetensors (timm MetaFormer key layout; without it the seeded synthetic
stand-in is used -- there is no network here to fetch deepghs/ccip_onnx), --batch, --device, --arch tiny (test geometry)."""
import argparse
import concurrent.futures
import datetime
import os
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


def encode_files(file_list, batch, cindex_encode, on_batch, size=384, gpu_resize=False, device=0):
    """The decode-ahead loop of gen_cfeatures.py:386-424: 8 threads prepare batch i+1 while the device encodes batch i.
    on_batch(paths_kept, features) is called per batch; failed loads are skipped (:392-395)."""
    import functools
    from hiptagsearch import cfeatures
    gen_image_ndarray = functools.partial(cfeatures.gen_image_ndarray, size=size, gpu_resize=gpu_resize, device=device)
    with concurrent.futures.ThreadPoolExecutor(max_workers=WORKER_NUM) as pool:
        nxt = pool.map(gen_image_ndarray, file_list[:batch])
        for s in range(0, len(file_list), batch):
            arrs = list(nxt)
            if s + batch < len(file_list):
                nxt = pool.map(gen_image_ndarray, file_list[s + batch: s + 2 * batch])
            keep = [(s + j, a) for j, a in enumerate(arrs) if a is not None]
            if keep:
                on_batch([i for i, _ in keep], cindex_encode([a for _, a in keep]))


def main(arg_str: list) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--dir', nargs=1, required=True, help='tagging target directory path')
    parser.add_argument('--after', nargs=1, help='tagging new images after this date (mtime attribute). Format: YYYY-MM-DD')
    parser.add_argument('--checkpoint', default=None)
    parser.add_argument('--batch', type=int, default=64)
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--workers', type=int, default=0,
                        help='decode / resize in this many processes (hiptagsearch/pipeline.py); the uint8 images go to the device u8 entry point')
    parser.add_argument('--operands', choices=['bf16', 'half'], default='half',
                        help='MFMA operand type of the encoder GEMMs (half: same matrix rate as bf16, 8x smaller activation rounding)')
    parser.add_argument('--gpu-resize', action='store_true',
                        help='decode threads only decode; the bilinear resize of gen_cfeatures.py:101 runs on the device (Pillow-exact kernel)')
    parser.add_argument('--gpu-jpeg', action='store_true',
                        help='with --workers: the worker processes only entropy-decode baseline JPEGs, the rest of the decode runs on the device')
    parser.add_argument('--arch', choices=['b36', 'tiny'], default='b36', help='b36: CAFormer-B36 widths @384 (the CCIP encoder); tiny: test geometry')
    args = parser.parse_args(arg_str)
    after_date = None
    if args.after is not None:
        try:
            after_date = datetime.datetime.strptime(args.after[0], '%Y-%m-%d').date()
        except Exception as e:
            print('%s: %s' % (type(e), str(e)))
            print('Invalid date format. format is YYYY-MM-DD')
            raise SystemExit(1)

    from hiptagsearch import dist as hdist
    dist, rank, world, device = hdist.init_from_env(args.device)          # before any GPU call
    import numpy as np
    from hiptagsearch import synth
    from hiptagsearch import cfeatures as cf
    from hiptagsearch.index import Similarity
    base = synth.CCIP_TINY if args.arch == 'tiny' else synth.CCIP_B36_384
    cfg = dict(base, operand_f16={'bf16': 0, 'half': 1}[args.operands])
    if args.checkpoint:
        encoder = cf.CCIPEncoder.from_safetensors(args.checkpoint, cfg, max_batch=args.batch, device=device)
    else:
        if rank == 0:
            print('no --checkpoint: using the seeded synthetic CCIP weights')
        encoder = cf.CCIPEncoder(cfg, synth.ccip_weights(cfg), max_batch=args.batch, device=device)
    feat_dim = encoder.out_dim

    # ---- rank 0: file list, backup, revision bookkeeping ----------------------------------------------------------
    file_list, save_name, cindex = None, INDEX_PREFIX, None
    if rank == 0:
        file_list = list_files_recursive(args.dir[0])
        print(f'{len(file_list)} files found')
        cindex = cf.CharacterFeatureIndex(encoder, device=device, prefix=INDEX_PREFIX)
        if feat_dim != 768:
            cindex.index = Similarity(INDEX_PREFIX, None, feat_dim, device)
        if after_date is not None:
            file_list = [p for p in file_list if datetime.date.fromtimestamp(os.path.getmtime(p)) >= after_date]
            print(f'{len(file_list)} files found after {after_date}')
            cf.backup_index_files('.')                                                  # :346-352
            max_number = cf.get_current_cfeature_number('.')                            # :354 (ValueError without an index, like the reference)
            print('copying index files to new index files')
            old = cf.CharacterFeatureIndex.load_latest(encoder, device, '.')            # :359-362 (+ csv alignment check)
            save_name = cf.revision_name(max_number + 1)
            cindex.index = Similarity(save_name, None, old.index.num_features, device, capacity=len(old.index) + len(file_list))
            cindex.index.add_matrix(old.index.matrix())                                 # :364-368 in one block
            cindex.paths = list(old.paths)
            if len(cindex.paths) != sum(1 for _ in open(INDEX_PREFIX + '.csv', encoding='utf-8')):
                with open(INDEX_PREFIX + '.csv', 'w', encoding='utf-8') as f:           # drop the surplus lines load_latest warned about
                    f.writelines(p + '\n' for p in cindex.paths)
            print('copying index files to new index files done')
    file_list = hdist.broadcast_object(file_list, dist)

    start = time.perf_counter()
    done = [0]

    def progress(k):
        done[0] += k
        el = time.perf_counter() - start
        print(f'{done[0]} files processed\n{el:.2f} seconds elapsed\n{el / max(done[0], 1):.4f} seconds per file\n', flush=True)

    def extract(arrs):                                                                  # :133-159
        if hasattr(arrs[0], "is_cuda"):                                                 # --gpu-resize: uint8 [S,S,3] device tensors
            import torch
            return np.asarray(encoder.forward_u8(torch.stack(list(arrs))), dtype=np.float32)
        return np.asarray(encoder(np.stack(arrs).astype(np.float32)), dtype=np.float32)

    if dist is not None:
        # ---- one process per GPU: contiguous blocks, one all-gather of feature rows --------------------------------
        import torch
        from hiptagsearch.shard import gather_rows, padded_rows_per_rank, shard_range
        n = len(file_list)
        lo, hi = shard_range(n, rank, world)
        per = padded_rows_per_rank(n, world)
        rows = np.full((max(per, 1), feat_dim + 1), np.nan, dtype=np.float32)           # column 0: 1 = encoded, NaN = padding / failed load

        def keep_rows(idx, feats):
            rows[idx, 0] = 1.0
            rows[idx, 1:] = feats
        encode_files(file_list[lo:hi], args.batch, extract, keep_rows, cfg['image_size'], args.gpu_resize, device)
        cdev = hdist.collective_device(dist, device)
        full = gather_rows(torch.from_numpy(rows[:per] if per else rows[:0]).to(cdev), n, dist).cpu().numpy()
        if rank == 0:
            ok = full[:, 0] == 1.0
            kept = [p for p, k in zip(file_list, ok) if k]
            with open(INDEX_PREFIX + '.csv', 'a', encoding='utf-8') as fcsv:
                for s in range(0, len(kept), 4096):
                    cindex.add_features(kept[s:s + 4096], full[ok][s:s + 4096, 1:])
                    fcsv.writelines(p + '\n' for p in kept[s:s + 4096])
            progress(len(kept))
            cindex.index.save(save_name)
        hdist.finish(dist)
        return

    with open(INDEX_PREFIX + '.csv', 'a', encoding='utf-8') as fcsv:

        def add(paths, feats):
            cindex.add_features(paths, feats)
            fcsv.writelines(p + '\n' for p in paths)                                    # :376,419
            fcsv.flush()
            progress(len(paths))
        if args.workers > 0:
            from hiptagsearch import pipeline
            with pipeline.DecodePool(args.workers, cfg["image_size"], args.batch, pipeline.CCIP, device_resize=args.gpu_resize or args.gpu_jpeg,
                                     device=device, device_jpeg=args.gpu_jpeg) as dpool:
                for kept, images in dpool.batches(file_list):
                    add(kept, encoder.forward_u8(images))                               # /255 and the CLIP normalisation on the device
        else:
            encode_files(file_list, args.batch, extract, lambda idx, feats: add([file_list[i] for i in idx], feats), cfg['image_size'],
                         args.gpu_resize, device)
    cindex.index.save(save_name)                                                        # :459


if __name__ == "__main__":
    main(sys.argv[1:])
