"""Drop-in for src/torch_core/infer.py -- see posthoc.py.   python -m vibo_amd.torch_core.infer <checkpoint.pth.tar>"""
from .posthoc import run


def main(argv=None):
    return run('infer', argv)


if __name__ == '__main__':
    main()
