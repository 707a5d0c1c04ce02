"""Drop-in `chooser.GPEIChooser`: same module name, same `init`/`next`, same
`chooser.GPEIChooser.pkl` state file as the reference module it shadows
(spearmint/spearmint/chooser/GPEIChooser.py); the EI grid runs on the GPU via libspx.so."""
from spearmint_amd import util as _util
from spearmint_amd.chooser import GPEIChooser as _impl


class GPEIChooser(_impl.GPEIChooser):
    # defined here so that self.__module__ == "chooser.GPEIChooser", which names the
    # state pickle exactly as the reference does (GPEIChooser.py: state_pkl)
    pass


def init(expt_dir, arg_string):
    return GPEIChooser(expt_dir, **_util.unpack_args(arg_string))
