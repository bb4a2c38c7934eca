#!/usr/bin/env python3
"""python tagging.py --dir D [--after YYYY-MM-DD]        (same flags as the reference, tagging.py:361-383)

Extra switches: --model vit-b16|eva02-l14, --checkpoint model.safetensors (timm key layout) and --labels
selected_tags.csv for a real wd tagger; without them the seeded synthetic stand-ins are used (no network here).
--compat reproduces the reference's dropped tail batch; --batch sets the device batch size; --precise trades about 5 % of the
throughput for the 1e-3 logit tolerance on flat / padded pictures (operand_f16 bit 4).
Input pipeline (hiptagsearch/pipeline.py): --workers N decodes in N processes through shared memory; --write-shards DIR
decodes the corpus once into packed uint8 shards and --shards DIR tags from them (utility/make_tensor_files.py's idea).

Multi-GPU (one process per GPU, RCCL all-gather of fixed-width tag rows restores file order):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tagging.py --dir D
writes the same tags-wd-tagger.txt as the single-process run (tests/test_gpu_multirank.py)."""
import argparse
import datetime
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(arg_str: list) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument('--dir', nargs=1, required=True, help='tagging target directory path')
    parser.add_argument('--after', nargs=1, help='tagging new images after this date (mtime attribute). Format: YYYY-MM-DD')
    parser.add_argument('--checkpoint', default=None)
    parser.add_argument('--labels', default=None)
    parser.add_argument('--compat', action='store_true')
    parser.add_argument('--model', choices=['vit-b16', 'eva02-l14', 'vit-tiny'], default='vit-b16',
                        help='vit-b16: wd-vit-tagger-v3 geometry (BASELINE.json contract model); eva02-l14: wd-eva02-large-tagger-v3, the repo tagging.py:45 names')
    parser.add_argument('--batch', type=int, default=64)
    parser.add_argument('--workers', type=int, default=0,
                        help='decode / resize in this many processes (shared-memory pipeline) instead of 8 threads')
    parser.add_argument('--write-shards', default=None, help='decode --dir once into packed uint8 shards in this directory and exit')
    parser.add_argument('--shards', default=None, help='tag the pre-decoded shards in this directory (written by --write-shards)')
    parser.add_argument('--gpu-resize', action='store_true',
                        help='decode threads (or the --workers processes) only decode; padding and the Resize(bicubic) of the transform run on the '
                             'device (Pillow-exact kernel)')
    parser.add_argument('--gpu-jpeg', action='store_true',
                        help='with --workers and --gpu-resize: the worker processes only entropy-decode baseline JPEGs; inverse DCT, chroma '
                             'upsampling and colour conversion (libjpeg-turbo\'s arithmetic, byte for byte) run on the device; other files as before')
    parser.add_argument('--synthetic', type=int, default=0, metavar='N',
                        help='tag N images of the synthetic benchmark corpus generated on the device (BASELINE.json configs[3]; --dir is ignored)')
    parser.add_argument('--precise', action='store_true',
                        help='attention output handed to the output projection as a hi | lo pair of 16-bit halves (operand_f16 |= 16): logits of '
                             'flat / padded pictures within 1e-3 of the fp32 reference at a trained checkpoint\'s scale, about 5 %% fewer images/s')
    parser.add_argument('--device', type=int, default=0)
    args = parser.parse_args(arg_str)
    # under torch.distributed.run (WORLD_SIZE > 1): one process per GPU, the process group comes up before any GPU call
    from hiptagsearch import dist as hdist
    dist, rank, world, device = hdist.init_from_env(args.device)
    from hiptagsearch.tagger import Predictor
    predictor = Predictor(device=device, max_batch=args.batch, compat=args.compat, gpu_resize=args.gpu_resize or args.gpu_jpeg, gpu_jpeg=args.gpu_jpeg,
                          precise=args.precise)
    from hiptagsearch import synth
    model_cfg = {'vit-b16': synth.VIT_B16_448, 'eva02-l14': synth.EVA02_L14_448, 'vit-tiny': synth.VIT_TINY}[args.model]   # vit-tiny: test geometry
    after_date = None
    if args.after is not None:
        try:
            after_date = datetime.datetime.strptime(args.after[0], '%Y-%m-%d').date()
        except Exception as e:
            print('%s: %s' % (type(e), str(e)))
            print('Invalid date format. format is YYYY-MM-DD')
            raise SystemExit(1)
    if args.write_shards:
        from hiptagsearch import pipeline
        cfg = model_cfg
        files = predictor.list_files_recursive(args.dir[0])
        if after_date is not None:
            files = predictor.filter_files_by_date(files, after_date)
        n = pipeline.write_shards(files, args.write_shards, cfg["image_size"], pipeline.TAGGER, args.workers or None)
        print(f'{n} of {len(files)} images written to {args.write_shards}')
        return
    predictor.load_model(args.checkpoint, args.labels, cfg=model_cfg)
    if dist is not None or args.synthetic:
        # every rank feeds itself: its block of the files (8 threads, or its own --workers pool), its slice of --shards, or its block of
        # the --synthetic corpus generated on its own GPU
        predictor.process_directory_sharded(args.dir[0], after_date, 10 if args.compat else args.batch, dist, rank, world,
                                            workers=args.workers, shards=args.shards, synthetic=args.synthetic)
        hdist.finish(dist)
        return
    predictor.process_directory(args.dir[0], after_date, batch_size=10 if args.compat else args.batch, workers=args.workers, shards=args.shards)


if __name__ == "__main__":
    main(sys.argv[1:])
