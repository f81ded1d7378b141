"""Drop-in for src/torch_core/predictives.py -- see posthoc.py.   python -m vibo_amd.torch_core.predictives <checkpoint.pth.tar>"""
from .posthoc import run


def main(argv=None):
    return run('predictives', argv)


if __name__ == '__main__':
    main()
