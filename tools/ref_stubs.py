"""In-memory stand-ins for the audio/GUI modules the reference imports at module level
(pyaudio, PyQt5, the Qt matplotlib backend).  They let ``tools/make_goldens.py`` import the
reference's DSP functions in a container that has neither an audio device nor Qt.  None of the
stubbed modules takes part in the arithmetic.  Used only by tools/; never shipped as product code."""
import sys, types, os
def install():
    def mod(name, **kw):
        m = types.ModuleType(name); m.__dict__.update(kw); sys.modules[name] = m; return m
    class _Any:
        def __init__(self,*a,**k): pass
        def __call__(self,*a,**k): return _Any()
        def __getattr__(self,n): return _Any()
    mod('pyaudio', paInt16=8, PyAudio=_Any)
    class QThread:
        def __init__(self,*a,**k): pass
    def pyqtSignal(*a,**k): return _Any()
    def pyqtSlot(*a,**k):
        return lambda f: f
    qtcore = mod('PyQt5.QtCore', QThread=QThread, QTimer=_Any, pyqtSignal=pyqtSignal, pyqtSlot=pyqtSlot, Qt=_Any(), QObject=object)
    class _W:
        def __init__(self,*a,**k): pass
    qtw = mod('PyQt5.QtWidgets', QMainWindow=_W, QApplication=_Any, QWidget=_W, QVBoxLayout=_Any, QHBoxLayout=_Any, QLabel=_Any, QPushButton=_Any, QDoubleSpinBox=_Any, QGroupBox=_Any, QGridLayout=_Any, QSizePolicy=_Any(), QMessageBox=_Any, QCheckBox=_Any, QSpinBox=_Any, QComboBox=_Any, QSlider=_Any, QFrame=_Any, QSplitter=_Any, QTabWidget=_Any, QStatusBar=_Any, QProgressBar=_Any)
    qtgl = mod('PyQt5.QtOpenGL')
    qtg = mod('PyQt5.QtGui', QFont=_Any, QColor=_Any, QPalette=_Any)
    mod('PyQt5', QtCore=qtcore, QtWidgets=qtw, QtOpenGL=qtgl, QtGui=qtg)
    import matplotlib; matplotlib.use('Agg')
    class FC:
        def __init__(self,*a,**k): pass
    mod('matplotlib.backends.backend_qtagg', FigureCanvasQTAgg=FC)
    mod('matplotlib.backends.backend_qt5agg', FigureCanvasQTAgg=FC)
