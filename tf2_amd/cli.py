"""The reference's host CLI on the MI355X path:

    python -m tf2_amd.cli <model_file> <q_file> <image_file> <verify_file> <num_images> [--net NET]

Same five positional arguments as Runtime_Engine/cnn/host/src/main.cpp:19-28.  ``--net`` names the
network config: a TF2_auto_config header (``resnet50.h``), an ``fpganetwork.bin``, or the builtin
``resnet50`` tables (default; identical to the shipped header, tests/test_config.py)."""
import argparse
import sys

from . import config as cfg, network


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="tf2_amd.cli", description=__doc__)
    for a in ("model_file", "q_file", "image_file", "verify_file"):
        ap.add_argument(a)
    ap.add_argument("num_images", type=int)
    ap.add_argument("--net", default="resnet50")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    print(f"model_file = {args.model_file}\nq_file = {args.q_file}\nimage_file  = {args.image_file}\n"
          f"verify_file_name = {args.verify_file}\nnum_images = {args.num_images}")
    tables = cfg.resnet50_tables() if args.net == "resnet50" else args.net
    net = network.NetWork(tables)
    net.Init(args.model_file, args.q_file, args.image_file, args.num_images, device=args.device)
    runner = network.Runner(None, net)
    runner.Init()
    out = runner.Run()
    print(f"Latency = {runner.latency_ms:.3f} ms\nThroughput = {runner.throughput_fps:.1f} fps")
    for i in range(args.num_images):
        try:
            err = network.Verify(i, args.verify_file, net.q, out, num_layer=net.num_layer)    # main.cpp:52
            print(f"Convolution {len(net.plan)} compare finished, error={err:f}")
        except OSError as e:
            print(f"verify file not readable: {e}")
        labels, probs = network.Evaluation(i, net.q, out, num_layer=net.num_layer)              # main.cpp:53
        for r, (l, p) in enumerate(zip(labels, probs)):
            print(f"rank={r}\tlabel={l:5d}\tprobability={p:f}")
    net.CleanUp()
    return 0


if __name__ == "__main__":
    sys.exit(main())
