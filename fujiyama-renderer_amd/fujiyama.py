"""Python-3 counterpart of the reference's scene-description emitter.

Mirrors the interface of the reference's tools/python_api/fujiyama.py:14-333
(`SceneInterface` with one method per scene command, `-R/--resolution`,
`-S/--pixelsamples`, `-P/--print` overrides applied at RenderScene time) so
that a scenes/*.py script only needs its `import fujiyama` line pointed here.
Like the reference it does no arithmetic: every call appends one text command
(grammar: SURVEY.md Appendix C, reference tools/scene_parser/command.cc:502-541)
and Run() hands the stream to the `scene` command parser -- here the in-process
C++ parser of libfjscene.so (fj_scene_run_text) instead of a `scene`
subprocess.
"""
import argparse
import os

# command -> number of arguments, tools/scene_parser/command.cc:502-541
COMMANDS = {
    "OpenPlugin": 2, "RenderScene": 1, "RunProcedure": 1, "SaveFrameBuffer": 2,
    "AddObjectToGroup": 2, "NewObjectInstance": 2, "NewFrameBuffer": 2,
    "NewObjectGroup": 1, "NewPointCloud": 1, "NewTurbulence": 1, "NewProcedure": 2,
    "NewRenderer": 1, "NewTexture": 2, "NewCamera": 2, "NewShader": 2, "NewVolume": 1,
    "NewCurve": 1, "NewLight": 2, "NewMesh": 1, "AssignFrameBuffer": 2,
    "AssignObjectGroup": 3, "AssignPointCloud": 3, "AssignTurbulence": 3,
    "AssignTexture": 3, "AssignVolume": 3, "AssignCamera": 2, "AssignShader": 3,
    "AssignCurve": 3, "AssignMesh": 3, "SetProperty1": 3, "SetProperty2": 4,
    "SetProperty3": 5, "SetProperty4": 6, "SetStringProperty": 3,
    "SetSampleProperty3": 6, "ShowPropertyList": 1,
}


def _fmt(v):
    if isinstance(v, float):
        return repr(v)
    return str(v)


class SceneInterface(object):
    def __init__(self, argv=None, parse_args=True):
        self.commands = []
        ap = argparse.ArgumentParser()
        ap.add_argument("-P", "--print", dest="p", action="store_true",
                        help="force to print scene descriptions instead of running")
        ap.add_argument("-R", "--resolution", dest="res", nargs=2, help="override resolution")
        ap.add_argument("-S", "--pixelsamples", dest="samples", nargs=2, help="override pixel samples")
        self.args = ap.parse_args(argv if argv is not None else ([] if not parse_args else None))

    # ---- generic emission: si.NewMesh('m') -> 'NewMesh m'
    def __getattr__(self, name):
        if name not in COMMANDS:
            raise AttributeError(name)
        arity = COMMANDS[name]

        def emit(*a):
            if len(a) != arity:
                raise TypeError("%s takes %d arguments (%d given)" % (name, arity, len(a)))
            self.commands.append(name + " " + " ".join(_fmt(x) for x in a))
        return emit

    def Comment(self, comment):
        self.commands.append("# %.128s" % comment)

    def OpenPlugin(self, name, plugin_path):
        # DSO extension is appended when missing (reference fujiyama.py:137-151)
        root, ext = os.path.splitext(plugin_path)
        path = plugin_path if ext == ".so" else plugin_path + ".so"
        self.commands.append("OpenPlugin %s %s" % (name, path))

    def RenderScene(self, renderer):
        if self.args.res:
            self.SetProperty2(renderer, "resolution", self.args.res[0], self.args.res[1])
        if self.args.samples:
            self.SetProperty2(renderer, "pixelsamples", self.args.samples[0], self.args.samples[1])
        self.commands.append("RenderScene %s" % renderer)

    def NewTexture(self, name, filename):
        # .mip as is; .hdr is converted to a .mip in a temp directory first, like the reference's
        # emitter does with its hdr2mip tool (reference fujiyama.py:222-236,335-347)
        root, ext = os.path.splitext(filename)
        if ext == ".hdr":
            import tempfile
            import uuid
            from . import host
            if not getattr(self, "tempdir", ""):
                self.tempdir = tempfile.mkdtemp()
            mip = os.path.join(self.tempdir, "%s_%s.mip" % (uuid.uuid4(), os.path.basename(root)))
            if host.lib().fj_hdr2mip(filename.encode(), mip.encode()) != 0:
                raise RuntimeError("hdr2mip failed for %s" % filename)
            filename = mip
        self.commands.append("NewTexture %s %s" % (name, filename))

    def SaveFrameBuffer(self, framebuffer, filename):
        self.commands.append("SaveFrameBuffer %s %s" % (framebuffer, filename))

    def text(self):
        return "\n".join(self.commands) + "\n"

    def Print(self):
        print(self.text(), end="")

    def Run(self):
        if self.args.p:
            self.Print()
            return 0
        from . import host
        return host.run_scene_text(self.text())
