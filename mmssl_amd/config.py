"""Process-wide flag namespace (`args`), the counterpart of the reference's module-level
`args = parse_args()` in Models.py:15 / main.py:34 / load_data.py:8 / batch_test.py:13.

Importing this package never parses sys.argv (library-friendly); `mmssl_amd.main`'s entry point
calls `configure(sys.argv[1:])`. Tests and embedding applications set attributes directly or
call `configure([...])`.
"""
from .utility.parser import parse_args

args = parse_args([])


def configure(argv=None, **overrides):
    """Re-parse flags IN PLACE (every module holds a reference to the same namespace)."""
    new = parse_args(argv if argv is not None else [])
    args.__dict__.clear()
    args.__dict__.update(new.__dict__)
    for k, v in overrides.items():
        setattr(args, k, v)
    return args


class HotCfg:
    """The live hot-path flags as plain attributes (what the sharded model / step read)."""

    def __init__(self, **over):
        a = args
        self.embed_size = a.embed_size
        self.head_num = a.head_num
        self.n_ui_layers = len(eval(a.weight_size))
        self.drop_rate = float(a.drop_rate)
        self.model_cat_rate = a.model_cat_rate
        self.id_cat_rate = a.id_cat_rate
        self.tau = a.tau
        self.cl_rate = a.cl_rate
        self.feat_reg_decay = a.feat_reg_decay
        self.decay = eval(a.regs)[0]
        self.batch_size = a.batch_size
        for k, v in over.items():
            setattr(self, k, v)
