"""Drop-in for src/torch_core/marginal.py -- see posthoc.py.   python -m vibo_amd.torch_core.marginal <checkpoint.pth.tar>"""
from .posthoc import run


def main(argv=None):
    return run('marginal', argv)


if __name__ == '__main__':
    main()
